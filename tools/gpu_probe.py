#!/usr/bin/env python3
"""Dev tool (GPU box): time the engine on bench-shaped batches per gap-length bracket."""
import os, sys, time, tempfile, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from figbird_amd import synth, api

def model_for(spec, seed=7):
    mc = synth.bench_model_case(seed, spec)
    d = tempfile.mkdtemp(prefix="figmodel_")
    p = synth.write_case(mc, d)
    m = api.model_from_files(p["scf"], p["tmp"], p["myout"], partial_flag=1 if spec.mode == "partial" else 0,
                             unmapped_flag=1 if spec.mode == "unmapped" else 0, script_itr=1, max_distance=spec.max_distance,
                             read_length=spec.read_len, neg_overlap=30, partial_len=mc.partial_len)
    return m, mc

if __name__ == "__main__":
    if os.environ.get("FIG_PROBE_TORCH"):          # A/B: does it matter that torch initialised the HIP runtime first (as in bench.py)?
        import torch
        torch.cuda.init(); torch.zeros(1, device="cuda")
    mode = sys.argv[1]
    # "mix": the default bench batch (GAGE gap mix, bench seed) of `reps` gaps; "mixN": the same recipe with N gaps
    lens = [int(x) if not x.startswith("mix") else -(int(x[3:]) if len(x) > 3 else 1) for x in sys.argv[2].split(",")]
    reps = int(sys.argv[3])
    rpg = float(sys.argv[4]) if len(sys.argv) > 4 else 1000.0
    spec = synth.BenchSpec(mode=mode) if mode == "unmapped" else synth.BenchSpec(mode="partial", read_len=101, insert_mean=180, insert_sd=10)
    spec.reads_per_gap_mean = rpg
    if mode == "partial" and len(sys.argv) > 4:
        spec.partial_cov = int(rpg)                  # partial mode: argv[4] = soft-clipped reads drawn per gap (bench.py's partial pass: 48)
    m, mc = model_for(spec)
    print("model", m.Tmin, m.Tmax, m.cutoff, m.stats, flush=True)
    eng = api.Engine(0, lib_path=os.environ.get("FIG_LIB"))
    eng.set_model(m)
    for G in lens:
        gl = np.full(reps, G) if G > 0 else None
        n_here = reps if G >= -1 else -G
        t0 = time.time(); batch, truth = synth.make_bench_batch(11 + G if G > 0 else 20260101, n_here, spec, gap_lengths=gl); tg = time.time() - t0
        print(f"[probe] batch G={G} n={n_here} starts at {time.time():.3f}", file=sys.stderr, flush=True)
        nr = int(batch.u_read_off[-1]) if mode == "unmapped" else int(batch.p_read_off[-1])
        t0 = time.time(); eng.upload(batch); tu = time.time() - t0
        t0 = time.time(); res = eng.fill_resident(); tf = time.time() - t0
        st = eng.stats(); eng.free_batch()
        mism = sum(sum(1 for a, b in zip(s, t.tobytes().decode()) if a != 'N' and a != b) for s, t in zip(res.strings, truth) if len(s) == len(t))
        filled = res.filled_bases
        reps_was, reps = reps, n_here
        print(json.dumps({"G": G, "gaps": reps, "reads_per_gap": nr / reps, "gen_s": round(tg, 2), "upload_s": round(tu, 3), "fill_s": round(tf, 3),
                          "kernel_ms": round(st["kernel_ms"], 2), "place_calls": st["place_calls"], "gflop": round(st["alg_flops"] / 1e9, 2), "spec_gflop": round(st["spec_flops"] / 1e9, 2),
                          "tflops": round(st["alg_flops"] / 1e12 / max(st["kernel_ms"] / 1e3, 1e-9), 3), "gaps_per_s": round(reps / max(st["kernel_ms"] / 1e3, 1e-9), 2),
                          "filled": filled, "mism": mism, "lens": [int(x) for x in res.filled_len[:4]]}), flush=True)
        reps = reps_was
