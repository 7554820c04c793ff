#!/usr/bin/env python3
"""Resource notes of every kernel in figbird_amd/lib/libfighip.so (llvm-readelf --notes on the gfx950 code object inside the
fat binary): VGPRs, spill counts, private segment, the figures VERDICT/DESIGN quote.  usage: python tools/kernel_notes.py [out.txt]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin/"

def main():
    out = sys.argv[1] if len(sys.argv) > 1 else None
    d = tempfile.mkdtemp()
    fat, co = os.path.join(d, "fat.bin"), os.path.join(d, "dev.co")
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", os.path.join(ROOT, "figbird_amd", "lib", "libfighip.so"), fat], check=True)
    subprocess.run([LLVM + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
    txt = subprocess.run([LLVM + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    rows, cur = [], {}
    for ln in txt.splitlines():
        m = re.match(r"\s+\.(name|vgpr_count|agpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|max_flat_workgroup_size):\s+(\S+)", ln)
        if not m:
            continue
        if m.group(1) == "name" and "name" in cur and len(cur) > 3:
            pass
        cur[m.group(1)] = m.group(2)
        if m.group(1) == "vgpr_spill_count":          # last key of a kernel's block (keys are sorted)
            rows.append(cur); cur = {}
    lines = ["# llvm-readelf --notes of the gfx950 code object in figbird_amd/lib/libfighip.so", "# kernel | vgpr | vgpr_spill_count | sgpr_spill_count | private_segment_fixed_size (B/lane) | max_flat_workgroup_size"]
    for r in rows:
        name = subprocess.run(["c++filt", r.get("name", "?")], capture_output=True, text=True).stdout.strip().split("(")[0]
        lines.append(f"{name} | {r.get('vgpr_count')} | {r.get('vgpr_spill_count')} | {r.get('sgpr_spill_count')} | {r.get('private_segment_fixed_size')} | {r.get('max_flat_workgroup_size')}")
    lines.append("# the spill counts and private segments are the register save frames of the engine's function calls (and control-code spills of the kernel bodies);")
    lines.append("# tools/isa_scratch_report.py (isa_scratch.txt) shows where the scratch instructions sit relative to the loops")
    s = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(s)
    print(s)

if __name__ == "__main__":
    main()
