#!/usr/bin/env python3
"""Where do the scratch_* instructions of the device code sit?  Compiles figbird_amd/csrc/fig_abi.hip to gfx950 assembly
and reports, per function, the number of scratch loads/stores by loop depth (from the compiler's own "Loop Header:
Depth=" annotations) -- the evidence behind DESIGN.md §4c's statement that the register save frames of the engine's
function calls lie outside every loop.  usage: python tools/isa_scratch_report.py [out.txt]"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else None
    d = tempfile.mkdtemp()
    s_path = os.path.join(d, "fig.s")
    err = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-Wno-unused-value", "--cuda-device-only", "-S",
                          "-o", s_path, os.path.join(ROOT, "figbird_amd", "csrc", "fig_abi.hip"), "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
    lines = open(s_path).read().splitlines()
    func, depth = None, 0
    per = collections.OrderedDict()
    for ln in lines:
        m = re.match(r"\s*\.type\s+(\S+),@function", ln)
        if m:
            func, depth = m.group(1), 0
            per[func] = collections.Counter()
            continue
        if func is None:
            continue
        if re.match(r"^\.LBB\d+_\d+:", ln):
            m2 = re.search(r"Depth=(\d+)", ln)
            depth = int(m2.group(1)) if m2 else 0
        elif "Depth=" in ln and ln.lstrip().startswith(";"):
            depth = int(re.search(r"Depth=(\d+)", ln).group(1))
        if re.search(r"\bscratch_(load|store)", ln):
            per[func][depth] += 1
        if re.match(r"^\.Lfunc_end", ln):
            func = None
    rep = ["# scratch_load/scratch_store instructions per function, by loop depth (0 = outside every loop)", ""]
    import subprocess as sp
    def dem(n):
        try:
            return sp.run(["c++filt", n], capture_output=True, text=True).stdout.strip()[:110]
        except Exception:
            return n
    tot_in_loops = 0
    for f, c in per.items():
        if not c:
            continue
        inl = sum(v for k, v in c.items() if k > 0)
        tot_in_loops += inl if ("fig_hot_estep" in f or "fig_hot_mle" in f) else 0
        rep.append(f"{dem(f):112s} " + "  ".join(f"depth{k}:{v}" for k, v in sorted(c.items())))
    rep += ["", f"scratch instructions inside loops of fig_hot_estep<...> / fig_hot_mle<...>: {tot_in_loops}", "",
            "# kernel resource usage (hipcc -Rpass-analysis=kernel-resource-usage)"]
    cur = None
    for ln in err.splitlines():
        m = re.search(r"Function Name: (\S+)", ln)
        if m:
            cur = dem(m.group(1)); continue
        m = re.search(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill|Occupancy \[waves/SIMD\]): (\d+)", ln)
        if m and cur:
            rep.append(f"{cur[:90]:92s} {m.group(1)} = {m.group(2)}")
    txt = "\n".join(rep) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt[-1500:])


if __name__ == "__main__":
    main()
