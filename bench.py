#!/usr/bin/env python3
"""bench.py -- gaps/s and filled-bases/s of the MI355X gap-fill engine on the synthetic 10^5-gap set of
BASELINE.json (SURVEY.md §8d recipe), with the FP64 roofline of the dominant kernel and the CPU baseline.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload: the jump-library (unmapped-mode) fill pass -- the expensive iteration of RunFigbird.sh's schedule --
over a batch of gaps drawn from the synthetic set's distribution: scaffolds of 50 kb with a gap every 5 kb,
GAGE-like gap-length mix, 2x150-bp reads with insert N(3500, 350), mean 10^3 reads per gap (10^8 reads /
10^5 gaps, capped at 3000 as Preprocess does), 0.5 % substitutions.  A bit-exact fill of the whole 10^5-gap
set is ~10^17 FP64 flops (hours on any hardware, CPU-years for the reference), so a "step" is one fill pass
over a fixed per-GPU batch of `--gaps-per-gpu` gaps sampled (seeded) from that distribution, resident in HBM.
Weak scaling: every rank gets its own batch of the same size; results are all-gathered (RCCL) every step.
One JSON line on stdout (rank 0)."""
from __future__ import annotations

import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X: 256 CUs x 4 SIMD x 16 FP64 lanes/clk x 2.4 GHz = 39.3 T FP64 instr-lanes/s.  The path cannot use FMA
# (the reference's x86 build rounds the multiply and the add separately, and filled bases must be bit-exact), so
# each instruction is ONE flop: peak = 39.3 TFLOP/s (the 78.6 TFLOP/s datasheet figure counts an FMA as two).
FP64_NOFMA_PEAK_TFLOPS = 39.3
HBM_PEAK_GBS = 8000.0


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=0)
    ap.add_argument("--gaps-per-gpu", type=int, default=512)
    ap.add_argument("--reads-per-gap", type=float, default=1000.0)
    ap.add_argument("--mix", default="gage", choices=["gage", "loguniform"])
    ap.add_argument("--mode", default="unmapped", choices=["unmapped", "partial"])
    ap.add_argument("--seed", type=int, default=20260101)
    ap.add_argument("--cpu-baseline", type=int, default=1)
    ap.add_argument("--partial-pass", type=int, default=1, help="also report the partial-mode pass (untimed extra)")
    ap.add_argument("--cpu-sample-gaps", type=int, default=0, help="0 = five gaps per host core from the >400-bp bracket")
    args = ap.parse_args()

    import torch
    from figbird_amd import api, synth, dist as fdist, build as fbuild

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        log(f"warning: WORLD_SIZE={world} != --gpus {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libfighip has no CPU path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    if args.mode == "unmapped":
        spec = synth.BenchSpec(mode="unmapped", reads_per_gap_mean=args.reads_per_gap, gap_mix=args.mix)
    else:
        spec = synth.BenchSpec(mode="partial", read_len=101, insert_mean=180, insert_sd=10, partial_cov=48, gap_mix=args.mix)

    # ---- run-level model: a small seeded myout.sam through the shipped host code (identical on every rank)
    work = tempfile.mkdtemp(prefix=f"figbench_r{rank}_")
    mc = synth.bench_model_case(7, spec)
    mp = synth.write_case(mc, os.path.join(work, "model"))
    model = api.model_from_files(mp["scf"], mp["tmp"], mp["myout"], partial_flag=int(spec.mode == "partial"),
                                 unmapped_flag=int(spec.mode == "unmapped"), script_itr=1, max_distance=spec.max_distance,
                                 read_length=spec.read_len, neg_overlap=30, partial_len=mc.partial_len)

    # ---- this rank's batch (weak scaling: same size and distribution on every rank, different seed)
    t0 = time.time()
    batch, truth = synth.make_bench_batch(args.seed + 1000 * rank, args.gaps_per_gpu, spec)
    n_gaps = batch.n_gaps
    n_reads = int(batch.u_read_off[-1]) if spec.mode == "unmapped" else int(batch.p_read_off[-1])
    log(f"[bench] rank {rank}: {n_gaps} gaps, {n_reads} reads generated in {time.time() - t0:.1f}s")

    eng = api.Engine(local)
    eng.set_model(model)
    eng.upload(batch)                        # inputs resident in HBM before the timed region
    up = eng.stats()

    def one_step():
        res = eng.fill_resident()
        ids = list(range(rank * n_gaps, (rank + 1) * n_gaps))
        if world > 1:
            fl, gt, ss = fdist.all_gather_results(ids, res.filled_len, res.gaptofill, res.strings, world * n_gaps, device=dev)
            filled = sum(len(s) - s.count("N") for s in ss)
        else:
            filled = res.filled_bases
        return res, filled, eng.stats()

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    flops = 0.0
    place_calls = 0
    filled = 0
    res = None
    for _ in range(args.steps):
        res, filled, st = one_step()
        kernel_ms += st["kernel_ms"]; flops += st["alg_flops"]; place_calls += st["place_calls"]
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        agg = torch.tensor([kernel_ms, flops, float(place_calls)], dtype=torch.float64, device=dev)
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        kernel_ms_sum, flops_all = float(agg[0]), float(agg[1])
        kernel_ms_avg = kernel_ms_sum / world
    else:
        kernel_ms_avg, flops_all = kernel_ms, flops

    total_gaps = world * n_gaps * args.steps
    gaps_per_s = total_gaps / elapsed
    out = {
        "metric": "gaps/sec (+ filled-bases/sec), synthetic 1e5-gap set recipe, jump-library fill pass",
        "value": gaps_per_s, "unit": "gaps/s", "n_gaps": world * n_gaps, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / max(args.steps, 1), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "filled_bases_per_s": filled * args.steps / elapsed,
        "config": {"workload": f"synthetic-1e5-gap recipe (SURVEY §8d), {spec.mode}-mode pass, batch of {n_gaps} gaps/GPU sampled from the {args.mix} gap mix",
                   "gaps_per_gpu": n_gaps, "reads_per_gap_mean": n_reads / max(n_gaps, 1), "read_len": spec.read_len,
                   "insert": [spec.insert_mean, spec.insert_sd], "substitution_rate": spec.err, "sharding": f"gaps x{world} ranks, 1 all-gather/step"},
    }
    # ---- roofline of the dominant kernel (fig_fill_kernel): algorithmic FP64 flops / HIP-event kernel time
    ksec = kernel_ms_avg / 1e3
    ach = flops_all / world / max(ksec, 1e-12) / 1e12 if world > 1 else flops_all / max(ksec, 1e-12) / 1e12
    alg_bytes = up["packed_bytes"] * args.steps + filled
    out["roofline"] = {"bound": "fp64_valu", "achieved": ach, "peak": FP64_NOFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_NOFMA_PEAK_TFLOPS,
                       "peak_note": "FP64 vector issue without FMA (256 CU x 4 SIMD x 16 lanes x 2.4 GHz): multiply and add must stay unfused for bit-exactness", "peak_with_fma": 2 * FP64_NOFMA_PEAK_TFLOPS, "frac_of_fma_peak": ach / (2 * FP64_NOFMA_PEAK_TFLOPS),
                       "traffic": None, "traffic_note": "not collectable inside this process; rocprofv3 --pmc passes of this command (profiles/round1): FETCH_SIZE x2 + WRITE_SIZE = 2.2e12 B per default step = 65 GB/s = 0.8 % of HBM peak, against 1.3e8 algorithmic bytes: per-workgroup scratch slabs, per-gap state slabs and register save frames of ~1800 resident workgroups cycling through L2, not input re-reads", "kernel": "fig_eval_kernel<LDS_TAB,NT> (+ fig_begin/replay/end: all launches of one fill, 3 class lanes)", "launches_per_step": st["n_launches"], "kernel_ms_per_step": kernel_ms_avg / max(args.steps, 1),
                       "alg_flops_per_step": flops_all / world / max(args.steps, 1), "placeReads_calls_per_step": place_calls / max(args.steps, 1),
                       "hbm": {"achieved": alg_bytes / max(ksec, 1e-12) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": alg_bytes / max(ksec, 1e-12) / 1e9 / HBM_PEAK_GBS, "alg_bytes_per_step": alg_bytes / max(args.steps, 1),
                               "note": "path is FP64-ALU bound (~1e5-1e7 flop/byte); HBM figure reported because the north star asks for it"}}

    # ---- CPU baseline: rank 0, N=1 only; bounded sample (the >400-bp bracket: one candidate length, a few EM
    # iterations, ~1-2 s per gap per core, five per core; a <=400-bp gap of this set costs 10^2-10^3 CPU-seconds)
    if rank == 0 and world == 1 and args.cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(args, spec, batch, mc, res, eng, work)
        except Exception as e:  # pragma: no cover
            out["cpu_baseline"] = {"value": None, "unit": "gaps/s", "cores": 0, "kind": "port", "sample": f"failed: {e!r}"}
    eng.free_batch()
    eng.close()
    # ---- the other mode of the reference's schedule (frag-library partial-mode pass over a batch of the same gap mix),
    # outside the timed region, reported beside the headline: rank 0, N=1 only
    if rank == 0 and world == 1 and args.mode == "unmapped" and args.partial_pass:
        try:
            out["partial_pass"] = partial_pass(args, local, work)
        except Exception as e:  # pragma: no cover
            out["partial_pass"] = {"value": None, "unit": "gaps/s", "note": f"failed: {e!r}"}
    shutil.rmtree(work, ignore_errors=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


def partial_pass(args, local, work):
    from figbird_amd import api, synth
    spec = synth.BenchSpec(mode="partial", read_len=101, insert_mean=180, insert_sd=10, partial_cov=48, gap_mix=args.mix)
    mc = synth.bench_model_case(7, spec)
    mp = synth.write_case(mc, os.path.join(work, "model_partial"))
    model = api.model_from_files(mp["scf"], mp["tmp"], mp["myout"], partial_flag=1, unmapped_flag=0, script_itr=1,
                                 max_distance=spec.max_distance, read_length=spec.read_len, neg_overlap=30, partial_len=mc.partial_len)
    batch, _ = synth.make_bench_batch(args.seed, args.gaps_per_gpu, spec)
    eng = api.Engine(local)
    eng.set_model(model)
    eng.upload(batch)
    res = eng.fill_resident()
    st = eng.stats()
    eng.free_batch(); eng.close()
    ksec = max(st["kernel_ms"] / 1e3, 1e-9)
    return {"value": batch.n_gaps / ksec, "unit": "gaps/s", "ms_per_step": st["kernel_ms"], "n_gaps": int(batch.n_gaps),
            "reads_per_gap_mean": float(batch.p_read_off[-1]) / max(batch.n_gaps, 1), "filled_bases_per_s": res.filled_bases / ksec,
            "achieved_tflops": st["alg_flops"] / ksec / 1e12, "frac_of_fp64_nofma_peak": st["alg_flops"] / ksec / 1e12 / FP64_NOFMA_PEAK_TFLOPS,
            "workload": "frag-library (2x101 bp, insert 180) partial-mode pass, same gap mix, kernel time of one fill"}


def cpu_baseline(args, spec, batch, mc, res, eng, work):
    from figbird_amd import synth, build as fbuild
    cores = min(os.cpu_count() or 1, 16)
    k = args.cpu_sample_gaps or cores * 5
    G = np.asarray(batch.gap_len)
    if spec.mode == "unmapped":
        nread = np.diff(batch.u_read_off)
        cand = [int(g) for g in np.argsort(nread) if G[g] > 400]
        label = ">400-bp bracket"
    else:
        cand = [int(g) for g in range(batch.n_gaps)]
        label = "all brackets"
    sample = cand[:k] if spec.mode == "unmapped" else cand[:max(k * 64, 256)]
    if not sample:
        raise RuntimeError("no gap fits the CPU sample")
    root = os.path.join(work, "cpu")
    paths = synth.write_batch_subset(batch, sample, mc, root, spec)
    order = paths["gap_order"]
    sel = [i for i, g in enumerate(order) if g in set(sample)]
    ref = os.path.join(fbuild.REFDIR, "Figbird.out")
    kind = "reference" if os.path.exists(ref) else "port"
    exe = [ref] if kind == "reference" else [fbuild.ORACLE, "figbird"]
    nproc = min(cores, len(sel))
    shards = [sel[i::nproc] for i in range(nproc)]
    with open(paths["tmp"] + "gaploads.txt", "w") as f:
        for sh in shards:
            f.write("".join(f"{g}\t" for g in sorted(sh)) + "\n")
    argv_tail = [paths["myout"], paths["tmp"], paths["gaps"], "30", str(mc.partial_len), "400", "0", str(int(spec.insert_mean))]
    t0 = time.perf_counter()
    procs = []
    for t, sh in enumerate(shards):
        cmd = exe + [paths["scf"], str(spec.max_distance), str(spec.read_len), "1", str(int(spec.mode == "partial")), str(int(spec.mode == "unmapped")),
                     str(t), str(len(sh))] + argv_tail
        procs.append(subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=root))
    for p in procs:
        p.wait()
    wall = time.perf_counter() - t0
    # parity spot-check of the sample against the GPU results + its algorithmic flops (GPU counters on the same gaps)
    ok = True
    for t, sh in enumerate(shards):
        lines = {int(l.split("\t")[0]): l.rstrip("\n").split("\t") for l in open(paths["tmp"] + f"gapout{t}.txt")}
        for i in sh:
            f = lines[i]
            g = order[i]
            ok &= (int(f[4]) == int(res.filled_len[g])) and ((f[5] if len(f) > 5 else "") == res.strings[g])
    sub = synth.subset_batch(batch, sample)
    eng.free_batch()
    eng.upload(sub)
    eng.fill_resident()
    st = eng.stats()
    eng.free_batch()
    eng.upload(batch)
    return {"value": len(sample) / wall, "unit": "gaps/s", "cores": nproc, "kind": kind,
            "sample": f"{len(sample)} gaps of the batch from the {label} ({int(G[sample].min())}-{int(G[sample].max())} bp, "
                      f"{int(np.diff(batch.u_read_off)[sample].mean()) if spec.mode == 'unmapped' else int(np.diff(batch.p_read_off)[sample].mean())} reads/gap), "
                      f"one {'oracle/_ref/Figbird.out (-O2 build of the reference)' if kind == 'reference' else 'oracle port'} process per core, {wall:.1f} s wall",
            "gflops": st["alg_flops"] / wall / 1e9, "gpu_same_sample_gaps_per_s": len(sample) / max(st["kernel_ms"] / 1e3, 1e-9),
            "gpu_same_sample_gflops": st["alg_flops"] / max(st["kernel_ms"] / 1e3, 1e-9) / 1e9, "parity_on_sample": bool(ok)}


if __name__ == "__main__":
    main()
