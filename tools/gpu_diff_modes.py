#!/usr/bin/env python3
"""Dev tool (GPU box): candidate traces of sequential vs candidate-parallel scheduling must be identical."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from figbird_amd import synth, api
from tools.gpu_probe import model_for

spec = synth.BenchSpec(mode="unmapped", reads_per_gap_mean=float(sys.argv[3]) if len(sys.argv) > 3 else 15.0)
m, mc = model_for(spec)
G = int(sys.argv[1]); n = int(sys.argv[2])
batch, truth = synth.make_bench_batch(11 + G, n, spec, gap_lengths=np.full(n, G))
out = {}
for mode in ("seq", "par"):
    if mode == "seq": os.environ["FIG_SCHED"] = "seq"
    else: os.environ.pop("FIG_SCHED", None)
    eng = api.Engine(0); eng.set_model(m)
    out[mode] = eng.fill(batch, debug_cand=1024); st = eng.stats(); eng.close()
    print(mode, st["place_calls"], st["alg_flops"], flush=True)
a, b = out["seq"], out["par"]
print("strings equal:", a.strings == b.strings)
for g in range(n):
    if a.n_place[g] != b.n_place[g]:
        print("gap", g, "n_place seq", a.n_place[g], "par", b.n_place[g], "ncand", len(a.cand[g]), "last", a.cand[g][-1], "best", max(a.cand[g], key=lambda c: c[3]))
for g in range(n):
    if a.cand[g] != b.cand[g]:
        print("gap", g, "n", len(a.cand[g]), len(b.cand[g]))
        for k, (x, y) in enumerate(zip(a.cand[g], b.cand[g])):
            if x != y: print("  first diff at cand", k, x, y); break
        break
