#!/usr/bin/env python3
"""GPU box: how good is dist.estimate_cost (what decides the multi-GPU shards and the in-class order)?  Fills the default
bench batch once with the per-gap placeReads counters on, takes  n_place[g] x R_g x min(G_g + L, 2200) x L  as the
measured cost of gap g (the E-step/MLE work of one placeReads call is R x W x L; W ~ G + L), and reports the correlation
with the estimate and the imbalance an LPT deal on the ESTIMATE leaves when weighed with the MEASURED cost."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from figbird_amd import api, synth, dist as fdist
from tools.gpu_probe import model_for

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    spec = synth.BenchSpec(mode="unmapped")
    m, mc = model_for(spec)
    batch, _ = synth.make_bench_batch(20260101, n, spec)
    eng = api.Engine(0); eng.set_model(m); eng.upload(batch)
    res = eng.fill_resident(debug_cand=1)
    eng.free_batch(); eng.close()
    G = np.asarray(batch.gap_len, dtype=np.float64); R = np.diff(batch.u_read_off).astype(np.float64); L = spec.read_len
    cand_len = np.where(G <= 133, 200.0, np.where(G <= 400, 1.5 * G, G))          # typical candidate length of the bracket
    measured = res.n_place.astype(np.float64) * R * np.minimum(cand_len + L, 2200.0) * L
    est = fdist.estimate_cost(batch.gap_len, R, L, True, mc.partial_len)
    out = {"n_gaps": n, "corr_log": float(np.corrcoef(np.log(est), np.log(measured + 1))[0, 1]),
           "ratio_p10_p50_p90": [float(x) for x in np.percentile(measured / est, [10, 50, 90])]}
    for world in (2, 4, 8):
        bins = fdist.partition_lpt(est, world)
        loads = np.array([measured[b].sum() for b in bins])
        out[f"imbalance_{world}"] = float(loads.max() / loads.mean())
    for lo, hi in [(0, 30), (31, 133), (134, 400), (401, 5000)]:
        k = (G >= lo) & (G <= hi)
        out[f"ratio_{lo}_{hi}"] = float(np.median(measured[k] / est[k])) if k.any() else None
        out[f"nplace_{lo}_{hi}"] = float(np.median(res.n_place[k])) if k.any() else None
    print(json.dumps(out))
