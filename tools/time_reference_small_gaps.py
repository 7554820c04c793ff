#!/usr/bin/env python3
"""Dev tool (CPU, this container): what the reference itself does per core on the bracket that dominates the bench step.

bench.py's cpu_baseline can only afford the >400-bp gaps inside the driver's run (one candidate length: ~1-2 s per gap and core);
96 % of a step's flops are <=400-bp gaps, each of which costs the reference 10^2-10^3 CPU-seconds.  This tool times
oracle/_ref/Figbird.out (the reference's own Figbird.cpp, -O2) on the committed bench-regime goldens of that bracket
(tests/golden/bench_{b25,b100,b160,cap40}: L = 150, insert 3500, 600-3000 reads, 270-320 candidate lengths), one process
per gap side by side, checks its gapout against the golden's, and divides the gap's algorithmic flops (the oracle's counter
= the device's, recorded in the golden's cands.json) by the time.  bench.py reads the result for its same-mix figure.

usage: python3 tools/time_reference_small_gaps.py <out.json> [names...]"""
import json, os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import util
from tools import build_test_infra as fbuild


def main():
    out = sys.argv[1]
    names = sys.argv[2:] or ["bench_b25", "bench_b100", "bench_b160", "bench_cap40"]
    ref = os.path.join(fbuild.REFDIR, "Figbird.out")
    assert os.path.exists(ref), "oracle/_ref/Figbird.out missing (make -C oracle)"
    base = tempfile.mkdtemp(prefix="figcpuref_")
    procs = []
    for nm in names:
        root = util.extract_golden(nm, os.path.join(base, nm))
        m = util.meta(root)
        for fn in ("gapout0.txt", "gaptofill0.txt", "draw0.txt"):
            p = os.path.join(root, "tmp", fn)
            if os.path.exists(p): os.remove(p)
        open(os.path.join(root, "tmp", "gaploads.txt"), "w").write("0\t\n")       # FillGaps.cpp:313-334 for one worker process holding gap 0
        t0 = time.time()
        procs.append((nm, root, m, t0, subprocess.Popen([ref] + m["figbird_argv"], cwd=root, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)))
    ends = {}
    while len(ends) < len(procs):                     # every process's own wall time (they run side by side on idle cores)
        for nm, root, m, t0, p in procs:
            if nm not in ends and p.poll() is not None:
                ends[nm] = (time.time() - t0, p.returncode)
                print(f"[done] {nm}: {ends[nm][0]:.1f} s rc {ends[nm][1]}", flush=True)
        time.sleep(0.25)
    res = []
    for nm, root, m, t0, p in procs:
        secs, rc = ends[nm]
        go = os.path.join(root, "tmp", "gapout0.txt")
        cands = json.load(open(os.path.join(root, "ref", "cands.json")))
        flops = float(cands["stats"][1]) if cands.get("stats") else None
        same = os.path.exists(go) and open(go).read() == open(os.path.join(root, "ref", "gapout0.txt")).read()
        res.append({"golden": nm, "gap_bp": m["g0"], "reads": m["n_reads"], "candidate_lengths": m["n_cands"], "rc": rc, "seconds": round(secs, 1),
                    "alg_flops": flops, "gflops_one_core": round(flops / secs / 1e9, 3) if flops else None, "gapout_equals_golden": bool(same)})
        print(res[-1], flush=True)
    ok = [r for r in res if r["gflops_one_core"]]
    rec = {"what": "oracle/_ref/Figbird.out (reference Figbird.cpp, g++ -O2) on bench-regime gaps of the <=400-bp bracket, one process per gap side by side on this container's cores",
           "host_cores": os.cpu_count(), "gaps": res,
           "gflops_one_core_flop_weighted": round(sum(r["alg_flops"] for r in ok) / sum(r["seconds"] for r in ok) / 1e9, 3) if ok else None}
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
