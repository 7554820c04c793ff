#!/usr/bin/env python3
"""bench.py -- gaps/s and filled-bases/s of the MI355X gap-fill engine on the synthetic 10^5-gap set of
BASELINE.json (SURVEY.md §8d recipe), with the FP64 roofline of the dominant kernel and the CPU baseline.

  python bench.py --gpus N --steps K --warmup W
  (N > 1 without a launcher: bench.py starts its own N ranks -- `python -m torch.distributed.run --nnodes=1
   --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same args>` as a child process, before anything touches the GPU,
   and relays the child's JSON line and exit code; under a launcher (WORLD_SIZE set) it is one of the ranks)

Workload: the jump-library (unmapped-mode) fill pass -- the expensive iteration of RunFigbird.sh's schedule --
over gaps drawn from the synthetic set's distribution: scaffolds of 50 kb with a gap every 5 kb, GAGE-like
gap-length mix, 2x150-bp reads with insert N(3500, 350), mean 10^3 reads per gap (10^8 reads / 10^5 gaps, capped
at 3000 as Preprocess does), 0.5 % substitutions.  A bit-exact fill of the whole 10^5-gap set is ~10^17 FP64
flops (hours on any hardware, CPU-years for the reference), so a "step" is one fill pass over a fixed seeded
sample of that distribution, resident in HBM (`full_set_seconds_est` extrapolates).

Multi-GPU: every rank generates the same global sample (same seed), the product's partitioner
(figbird_amd.dist.partition_lpt on estimate_cost: the role of FillGaps.cpp:456-649) deals it into N shards, each rank
fills its shard through the C ABI, and one all-gather of packed byte buffers per step reassembles the results.
`--scaling strong` (default): ONE fixed seeded set of `--total-gaps` gaps (default 8 x `--gaps-per-gpu` = 4096) whatever N is
-- BASELINE's metric is one set filled at 1/2/4/8 GPUs, so N = 1 fills all 4096 gaps per step and N = 8 fills 512 per rank;
`--scaling weak`: the sample is `--gaps-per-gpu` x N gaps.

Wall budget: the whole run (imports, generation, warm-up and timed steps, then the CPU baseline, the partial pass and the
per-bracket probes) is kept inside `--budget-s` seconds.  The first fill is timed; warm-up and step counts are then clamped
to what fits (with a reserve for the extras that follow the timed loop) and the line reports the counts actually run
(`steps`, `warmup`) beside `requested_steps` / `requested_warmup`.
One JSON line on stdout (rank 0), also when the budget runs out or the process receives SIGTERM."""
from __future__ import annotations

import os
import time

T_PROC0 = time.perf_counter()
# one hardware queue per scheduler lane (up to nine streams per fill; the runtime's default is 4): must be in the environment
# before the HIP runtime starts, i.e. before torch is imported -- see fig_ctx_create
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import argparse
import json
import os
import shutil
import signal
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X: 256 CUs x 4 SIMD x 16 FP64 lanes/clk x 2.4 GHz = 39.3 T FP64 instr-lanes/s.  The path cannot use FMA
# (the reference's x86 build rounds the multiply and the add separately, and filled bases must be bit-exact), so
# each instruction is ONE flop: peak = 39.3 TFLOP/s (the 78.6 TFLOP/s datasheet figure counts an FMA as two).
FP64_NOFMA_PEAK_TFLOPS = 39.3
HBM_PEAK_GBS = 8000.0
FULL_SET_GAPS = 100000
TRAFFIC_SIDECAR = os.path.join(ROOT, "profiles", "traffic_sidecar.json")

OUT = {}                 # the JSON line, filled in as results arrive
_EMITTED = False


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.perf_counter() - T_PROC0:6.1f}s]", *a, file=sys.stderr, flush=True)


def emit():
    global _EMITTED
    if _EMITTED or int(os.environ.get("RANK", "0")) != 0 or not OUT:
        return
    _EMITTED = True
    print(json.dumps(OUT), flush=True)


def _on_term(signum, frame):          # killed from outside: leave whatever has been measured so far
    OUT.setdefault("note", f"terminated by signal {signum} before the run finished; fields present are final")
    emit()
    os._exit(0 if "value" in OUT else 1)


def launch_ranks(n, argv):
    """`python bench.py --gpus N` with no launcher: run the N ranks as a child `torch.distributed.run` (one process per GPU,
    rendezvous on 127.0.0.1), relay rank 0's JSON line, return the child's exit code.  The reference starts its own workers
    the same way (FillGaps.cpp:668-679)."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "4")
    env["FIGBENCH_T0_OFFSET"] = f"{time.perf_counter() - T_PROC0:.3f}"
    print(f"[bench] no launcher: starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)

    def _fwd(signum, frame):
        child.send_signal(signum)
    signal.signal(signal.SIGTERM, _fwd)
    got = False
    for ln in child.stdout:
        if ln.lstrip().startswith("{"):
            got = True
        sys.stdout.write(ln); sys.stdout.flush()
    rc = child.wait()
    if rc != 0 or not got:
        print(f"[bench] the {n}-rank child exited with code {rc}{'' if got else ' and printed no JSON line'}", file=sys.stderr, flush=True)
        return rc or 1
    return 0


def n_set_of(args, world):
    return (args.total_gaps or 8 * args.gaps_per_gpu) if args.scaling == "strong" else args.gaps_per_gpu * world


def workload_key(args, world):
    return f"{args.mode}|{args.mix}|set{n_set_of(args, world)}|r{args.reads_per_gap:g}|s{args.seed}|n{world}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--budget-s", type=float, default=float(os.environ.get("FIGBENCH_BUDGET_S", "540")),
                    help="wall budget of the whole run; warm-up/steps are clamped to fit")
    ap.add_argument("--gaps-per-gpu", type=int, default=512)
    ap.add_argument("--reads-per-gap", type=float, default=1000.0)
    ap.add_argument("--mix", default="gage", choices=["gage", "loguniform"])
    ap.add_argument("--mode", default="unmapped", choices=["unmapped", "partial"])
    ap.add_argument("--seed", type=int, default=20260101)
    ap.add_argument("--cpu-baseline", type=int, default=1)
    ap.add_argument("--partial-pass", type=int, default=1, help="also report the partial-mode pass (untimed extra)")
    ap.add_argument("--cpu-sample-gaps", type=int, default=0, help="0 = five gaps per host core from the >400-bp bracket")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"], help="strong: one fixed set of --total-gaps for every N; weak: gaps-per-gpu x N gaps")
    ap.add_argument("--total-gaps", type=int, default=0, help="size of the fixed set of --scaling strong (0 = 8 x gaps-per-gpu)")
    ap.add_argument("--bracket-probes", type=int, default=1, help="measure the cheap per-bracket probes inside this run (N=1 only)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # No launcher around us: start the N ranks ourselves.  Decided before torch / HIP are touched in this process, which
        # stays a plain parent (it never initialises the GPU, the ranks are fresh child processes, nothing is exec'ed).
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    signal.signal(signal.SIGTERM, _on_term)

    import torch
    from figbird_amd import api, synth, dist as fdist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libfighip has no CPU path")
    # Rehearsal knobs for a box with fewer GPUs than ranks (never set by the driver): FIGBENCH_DEVICE=<n> puts every rank on
    # GPU n, FIGBENCH_BACKEND=gloo moves the collectives to host tensors (RCCL refuses two ranks on one device).
    backend = os.environ.get("FIGBENCH_BACKEND", "nccl")
    if "FIGBENCH_DEVICE" in os.environ:
        local = int(os.environ["FIGBENCH_DEVICE"])
    torch.cuda.set_device(local)
    gdev = torch.device("cuda", local)
    dev = gdev if backend == "nccl" else torch.device("cpu")      # where the collectives' tensors live
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=gdev)
        else:
            dist.init_process_group(backend)
        world = dist.get_world_size()          # n_gpus of the line = the ranks the process group reports
        warm = torch.ones(1, device=dev)
        dist.all_reduce(warm)                  # communicator set-up (RCCL builds its rings on the first collective) outside every timed region
        assert int(warm.item()) == world
    log(f"imports + init done (budget {args.budget_s:.0f} s)")

    t_parent = float(os.environ.get("FIGBENCH_T0_OFFSET", "0")) + (8.0 if "FIGBENCH_T0_OFFSET" in os.environ else 0.0)   # self-launched: parent + launcher start-up

    def left():
        return args.budget_s - t_parent - (time.perf_counter() - T_PROC0)

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

    # ---- the global sample (identical on every rank) and this rank's shard of it
    n_set = n_set_of(args, world)
    gbatch, truth = synth.make_bench_batch(args.seed, n_set, spec)
    n_global = gbatch.n_gaps
    off = gbatch.u_read_off if spec.mode == "unmapped" else gbatch.p_read_off
    cost = fdist.estimate_cost(np.asarray(gbatch.gap_len), np.diff(off), spec.read_len, spec.mode == "unmapped", mc.partial_len)
    shards = fdist.partition_lpt(cost, world)
    my_ids = shards[rank]
    batch = gbatch if world == 1 else synth.subset_batch(gbatch, my_ids)
    n_gaps = batch.n_gaps
    n_reads = int(batch.u_read_off[-1]) if spec.mode == "unmapped" else int(batch.p_read_off[-1])
    log(f"rank {rank}: shard of {n_gaps}/{n_global} gaps, {n_reads} reads")

    eng = api.Engine(local)
    eng.set_model(model)
    eng.upload(batch)                        # inputs resident in HBM before the timed region
    up = eng.stats()

    def one_step():
        res = eng.fill_resident()
        if world > 1:
            fl, gt, ss = fdist.all_gather_packed(my_ids, res, n_global, device=dev)
            filled = ss.filled_bases()
        else:
            filled = res.filled_bases
        return res, filled, eng.stats()

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def allmax(x):
        if world == 1:
            return float(x)
        import torch.distributed as dist
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    OUT.update({
        "metric": "gaps/sec (+ filled-bases/sec), synthetic 1e5-gap set recipe, jump-library fill pass",
        "unit": "gaps/s", "n_gaps": n_global, "n_gpus": world, "requested_steps": args.steps, "requested_warmup": args.warmup,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"synthetic-1e5-gap recipe (SURVEY §8d), {spec.mode}-mode pass, seeded sample of "
                               + (f"{args.gaps_per_gpu} gaps/GPU" if args.scaling == "weak" else f"{n_global} gaps in all (fixed set)") + f" from the {args.mix} gap mix",
                   "workload_key": workload_key(args, world), "gaps_per_gpu": n_global / world, "reads_per_gap_mean": float(np.diff(off).mean()), "read_len": spec.read_len,
                   "insert": [spec.insert_mean, spec.insert_sd], "substitution_rate": spec.err,
                   "sharding": f"{n_global} gaps dealt LPT on estimated cost over {world} rank(s), no data-path collective, 1 all-gather of packed results per step"},
    })

    # ---- first fill: timed on its own, counts as warm-up; everything after it is clamped to the budget
    # (--warmup 0: no first fill and no clamping -- exactly --steps fills, for the profiler passes of tools/profile_bench.sh)
    barrier()
    if args.warmup > 0:
        t0 = time.perf_counter()
        res, filled, st = one_step()
        barrier()
        t_first = allmax(time.perf_counter() - t0)
        warm_done = 1
        log(f"first fill {t_first:.1f} s ({st['kernel_ms'] / 1e3:.1f} s of kernels), {left():.0f} s of budget left")
    else:
        t_first, warm_done = 0.0, 0

    # ---- clamp warm-up and steps to what is left (10 s reserve for teardown); the timed steps come before the extras
    # N = 1 keeps a reserve for the extras that follow the timed loop (CPU baseline ~60 s, partial pass ~10 s, bracket probes ~35 s)
    extras_reserve = 110.0 if (world == 1 and (args.cpu_baseline or args.partial_pass or args.bracket_probes)) else 0.0
    left_min = -allmax(-left())                       # the rank with the least budget left decides
    fit = int(max(0.0, left_min - 10.0 - extras_reserve) / max(t_first, 1e-3)) if warm_done else args.steps
    steps = max(1, min(args.steps, fit))
    warm_more = max(0, min(args.warmup - warm_done, fit - steps))
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([steps, warm_more], dtype=torch.int64, device=dev)
        dist.broadcast(t, src=0)
        steps, warm_more = int(t[0]), int(t[1])
    log(f"running {warm_more} more warm-up + {steps} timed steps (requested {args.warmup}/{args.steps})")
    for _ in range(warm_more):
        one_step()
    barrier()
    t0 = time.perf_counter()
    kernel_ms = 0.0
    flops = 0.0
    place_calls = 0
    for i in range(steps):
        res, filled, st = one_step()
        kernel_ms += st["kernel_ms"]; flops += st["alg_flops"]; place_calls += st["place_calls"]
        log(f"step {i + 1}/{steps}: kernels {st['kernel_ms'] / 1e3:.2f} s")
    barrier()
    elapsed = allmax(time.perf_counter() - t0)
    per_rank_kernel_ms = [kernel_ms / steps]
    if world > 1:
        import torch.distributed as dist
        agg = torch.tensor([kernel_ms, flops, float(place_calls)], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(agg) for _ in range(world)]
        dist.all_gather(every, agg)
        per_rank_kernel_ms = [float(e[0]) / steps for e in every]
        flops_all = float(sum(float(e[1]) for e in every))
        place_all = float(sum(float(e[2]) for e in every))
        kernel_ms_max = max(float(e[0]) for e in every)
    else:
        flops_all, place_all, kernel_ms_max = flops, float(place_calls), kernel_ms

    total_gaps = n_global * steps
    gaps_per_s = total_gaps / elapsed
    OUT.update({"value": gaps_per_s, "steps": steps, "warmup": warm_done + warm_more, "ms_per_step": 1e3 * elapsed / steps,
                "filled_bases_per_s": filled * steps / elapsed,
                "full_set_seconds_est": FULL_SET_GAPS / gaps_per_s,
                "full_set_note": f"{FULL_SET_GAPS} gaps of this mix at the measured rate; the timed step is a {n_global}-gap seeded sample of that set",
                "first_fill_s": t_first, "wall_s_before_timed_loop": t0 - T_PROC0})
    # ---- roofline of the dominant kernel (fig_eval_kernel): algorithmic FP64 flops / HIP-event kernel time.  For N > 1
    # the chip-level figure is per GPU: this rank set's flops / N over the slowest rank's kernel time.
    ksec = kernel_ms_max / 1e3
    ach = flops_all / world / max(ksec, 1e-12) / 1e12
    alg_bytes = up["packed_bytes"] * steps + filled * steps / max(world, 1)
    traffic, traffic_note = read_traffic(args, world)
    OUT["roofline"] = {
        "bound": "fp64_valu", "achieved": ach, "peak": FP64_NOFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_NOFMA_PEAK_TFLOPS,
        "peak_note": "FP64 vector issue without FMA (256 CU x 4 SIMD x 16 lanes x 2.4 GHz): multiply and add must stay unfused for bit-exactness",
        "peak_with_fma": 2 * FP64_NOFMA_PEAK_TFLOPS, "frac_of_fma_peak": ach / (2 * FP64_NOFMA_PEAK_TFLOPS),
        "executed_note": "achieved counts the SURVEY §8d formula (4L E-step + 1L MLE + pile-up adds per placement); the MLE term is pruned exactly on the device, see executed_frac",
        "traffic": traffic, "traffic_note": traffic_note,
        "kernel": "fig_eval_kernel<LDS_TAB,NT> (+ fig_begin/replay/end: all launches of one fill, concurrent class lanes)",
        "launches_per_step": st["n_launches"], "kernel_ms_per_step": kernel_ms_max / steps, "per_rank_kernel_ms_per_step": per_rank_kernel_ms,
        "alg_flops_per_step": flops_all / world / steps, "placeReads_calls_per_step": place_all / steps,
        "hbm": {"achieved": alg_bytes / max(ksec, 1e-12) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": alg_bytes / max(ksec, 1e-12) / 1e9 / HBM_PEAK_GBS, "alg_bytes_per_step": alg_bytes / steps,
                "note": "path is FP64-ALU bound (~1e5-1e7 flop/byte); HBM figure reported because the north star asks for it"}}
    if st.get("spec_flops", 0) > 0 and st.get("mle_alg_flops", 0) > 0:
        # credited flops the device really executed: everything but the pruned part of the MLE term (ratios taken over all
        # evaluations of the last step, discarded speculation included)
        mle_share = st["mle_alg_flops"] / st["spec_flops"]
        mle_done = min(1.0, st["mle_exec_flops"] / st["mle_alg_flops"])
        OUT["roofline"]["mle_share_of_alg_flops"] = mle_share
        OUT["roofline"]["mle_fraction_executed"] = mle_done
        OUT["roofline"]["executed_frac"] = OUT["roofline"]["frac"] * (1.0 - mle_share * (1.0 - mle_done))
        OUT["roofline"]["discarded_speculation_frac"] = max(0.0, 1.0 - st["alg_flops"] / st["spec_flops"])
    # ---- CPU baseline, the other mode's pass and the per-bracket probes (rank 0, N=1 only): after the timed loop, inside what is left of the budget
    extras_s = 0.0
    if rank == 0 and world == 1 and args.cpu_baseline:
        if left() > 45:
            t = time.perf_counter()
            try:
                OUT["cpu_baseline"] = cpu_baseline(args, spec, batch, mc, res, eng, work, st, min(left() - 20, 110.0))
            except Exception as e:  # pragma: no cover
                OUT["cpu_baseline"] = {"value": None, "unit": "gaps/s", "cores": 0, "kind": "port", "sample": f"failed: {e!r}"}
            extras_s += time.perf_counter() - t
            log(f"cpu_baseline done in {time.perf_counter() - t:.1f} s")
        else:
            OUT["cpu_baseline"] = {"value": None, "unit": "gaps/s", "cores": 0, "kind": "reference", "sample": "skipped: wall budget too short"}
    if rank == 0 and world == 1 and args.mode == "unmapped" and args.partial_pass and left() > 25:
        t = time.perf_counter()
        try:
            OUT["partial_pass"] = partial_pass(args, local, work)
        except Exception as e:  # pragma: no cover
            OUT["partial_pass"] = {"value": None, "unit": "gaps/s", "note": f"failed: {e!r}"}
        extras_s += time.perf_counter() - t

    if rank == 0 and world == 1 and args.mode == "unmapped" and args.bracket_probes and left() > 30:
        t = time.perf_counter()
        try:
            OUT["per_bracket"] = bracket_probes(args, eng, batch, spec, min(45.0, left() - 12))
        except Exception as e:  # pragma: no cover
            OUT["per_bracket"] = {"note": f"failed: {e!r}"}
        extras_s += time.perf_counter() - t
        log(f"bracket probes done in {time.perf_counter() - t:.1f} s")

    if "cpu_baseline" in OUT and OUT["cpu_baseline"].get("gflops"):
        cb = OUT["cpu_baseline"]
        # same-mix figure: the reference does the same algorithmic flops per gap (control flow is bit-identical), so the
        # mix costs it alg_flops_per_step / its measured flop rate
        mix_s = (flops_all / steps) / (cb["gflops"] * 1e9)
        cb["value"] = n_global / mix_s
        cb["value_note"] = ("same-mix extrapolation: this step's algorithmic flops / the reference's measured flop rate on the sample "
                            f"= {mix_s:.0f} s per step on {cb['cores']} cores; measured_sample_gaps_per_s is the raw sample figure (cheapest bracket only)")
        # the <=400-bp bracket holds ~96 % of the step's flops and cannot be sampled inside this run (10^2-10^3 CPU-seconds per gap):
        # its per-core flop rate comes from the committed offline measurement (tools/time_reference_small_gaps.py), when there is one
        small = read_small_gap_rate()
        if small and cb.get("kind") == "reference":
            G_all = np.asarray(gbatch.gap_len)
            share_gt = float(cost[G_all > 400].sum() / max(cost.sum(), 1e-30)) if spec.mode == "unmapped" else 0.0
            rate_small = small["gflops_one_core_flop_weighted"] * cb["cores"] * 1e9
            mix2 = (flops_all / steps) * (share_gt / (cb["gflops"] * 1e9) + (1.0 - share_gt) / rate_small)
            cb["value_sampled_bracket_only"] = cb["value"]
            cb["value"] = n_global / mix2
            cb["dominant_bracket"] = {"gflops_one_core": small["gflops_one_core_flop_weighted"], "source": small["source"], "gaps": small["gaps"],
                                      "flop_share_of_gt400bp_gaps_est": share_gt}
            cb["value_note"] = (f"same-mix extrapolation: this step's algorithmic flops, {100 * (1 - share_gt):.0f} % of them priced at the reference's flop rate on "
                                f"<=400-bp bench-regime gaps ({small['gflops_one_core_flop_weighted']:.2f} GFLOP/s per core, measured offline on {len(small['gaps'])} gaps, x {cb['cores']} cores) "
                                f"and the rest at the rate measured in this run on the >400-bp sample = {mix2:.0f} s per step; "
                                "measured_sample_gaps_per_s is the raw sample figure (cheapest bracket only)")
    eng.free_batch()
    eng.close()
    shutil.rmtree(work, ignore_errors=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
    if rank == 0 and args.mode == "unmapped":
        OUT.setdefault("per_bracket", {})["from_file"] = read_brackets()
    OUT["fills_run"] = warm_done + warm_more + steps        # fills of the main batch (what a profiler around this command sees)
    OUT["wall_s_total"] = time.perf_counter() - T_PROC0
    emit()


def bracket_probes(args, eng, main_batch, spec, budget_s):
    """Per-bracket figures measured in THIS run (gaps/s per bracket beside the blended number): batches of equal
    gaps at ~1000 reads each, kernel time of one fill on the resident batch.  Only the cheap brackets fit the wall budget
    (one-candidate gaps > 400 bp, and a 64-gap probe of the <= 400-bp regime); the rest is read from the committed probe."""
    from figbird_amd import synth
    t0 = time.perf_counter()
    rows = []
    eng.free_batch()
    try:
        for G, n in ((800, 256), (1500, 256), (1800, 256), (2000, 256), (100, 64)):
            if time.perf_counter() - t0 > budget_s * (0.6 if G > 400 else 0.35):
                rows.append({"gap_bp": G, "note": "skipped: probe budget spent"})
                continue
            b, _ = synth.make_bench_batch(args.seed + G, n, synth.BenchSpec(mode="unmapped", reads_per_gap_mean=1000.0), gap_lengths=np.full(n, G))
            eng.upload(b)
            eng.fill_resident()
            st = eng.stats()
            eng.free_batch()
            ks = max(st["kernel_ms"] / 1e3, 1e-9)
            rows.append({"gap_bp": G, "n_gaps": n, "reads_per_gap": float(b.u_read_off[-1]) / n, "gaps_per_s": n / ks, "kernel_s": ks,
                         "tflops": st["alg_flops"] / ks / 1e12, "frac_of_fp64_nofma_peak": st["alg_flops"] / ks / 1e12 / FP64_NOFMA_PEAK_TFLOPS})
    finally:
        eng.upload(main_batch)
    return {"measured_in_this_run": rows, "note": "batches of equal gaps, ~1000 reads each, kernel time of one fill; 256 gaps fill every CU once for the one-candidate brackets; the 64-gap 100-bp probe runs 64 gaps x ~270 candidates through the scheduler"}


def read_brackets():
    """Per-bracket figures (SURVEY §8d asks for gaps/s per bracket beside the blended number): the batches of 512 equal gaps
    are too long to run inside this command's budget, so the line carries the latest committed probe of the same build
    (tools/gpu_probe.py -> profiles/round3/probe512_final.txt), labelled as such."""
    path = os.path.join(ROOT, "profiles", "round3", "probe512_final.txt")
    try:
        rows = []
        for ln in open(path):
            if ln.startswith("{"):
                d = json.loads(ln)
                rows.append({"gap_bp": d["G"], "gaps_per_s": d["gaps_per_s"], "tflops": d["tflops"], "frac_of_fp64_nofma_peak": round(d["tflops"] / FP64_NOFMA_PEAK_TFLOPS, 4)})
        return {"source": "profiles/round3/probe512_final.txt (tools/gpu_probe.py unmapped <G> 512: 512 equal gaps, ~1000 reads each, kernel time of one fill; not measured in this run)", "unmapped": rows}
    except Exception as e:
        return {"source": f"unavailable: {e!r}"}


def csrc_sha():
    """Content hash of the kernel sources (figbird_amd/csrc/*.h, *.hip): ties profiles/traffic_sidecar.json to the build it was
    measured on without needing git (the GPU boxes have no .git)."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "figbird_amd", "csrc")
    for fn in sorted(os.listdir(d)):
        if fn.endswith((".h", ".hip")):
            h.update(fn.encode()); h.update(open(os.path.join(d, fn), "rb").read())
    return h.hexdigest()[:12]


def read_small_gap_rate():
    """The reference's own flop rate per core on bench-regime gaps of the <=400-bp bracket, measured offline
    (tools/time_reference_small_gaps.py -> profiles/round4/cpu_reference_le400bp_gaps.json): oracle/_ref/Figbird.out on the
    committed bench goldens, outputs equal to the goldens'."""
    path = os.path.join(ROOT, "profiles", "round4", "cpu_reference_le400bp_gaps.json")
    try:
        d = json.load(open(path))
        gaps = [{k: g[k] for k in ("golden", "gap_bp", "reads", "candidate_lengths", "seconds", "gflops_one_core")} for g in d["gaps"] if g.get("rc") == 0 and g.get("gapout_equals_golden")]
        if not gaps or not d.get("gflops_one_core_flop_weighted"):
            return None
        return {"gflops_one_core_flop_weighted": float(d["gflops_one_core_flop_weighted"]), "gaps": gaps, "source": "profiles/round4/cpu_reference_le400bp_gaps.json (not measured in this run)"}
    except Exception:
        return None


def read_traffic(args, world):
    """HBM bytes per step AND PER GPU (like roofline.achieved / alg_flops_per_step) from the PMC passes of this same command at
    N = 1 (tools/profile_bench.sh writes the sidecar from `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs; counters
    cannot be read from inside the process).  N > 1 is scaled from the 1-GPU pass: no PMC pass is run on the multi-GPU node."""
    try:
        sc = json.load(open(TRAFFIC_SIDECAR))
        src = f"(source: {os.path.relpath(TRAFFIC_SIDECAR, ROOT)}, head {sc.get('head', '?')}"
        stale = "; STALE: the PMC passes were recorded with other kernel sources (csrc hash differs)" if sc.get("csrc_sha") and sc.get("csrc_sha") != csrc_sha() else ""
        ent = sc.get(workload_key(args, 1))
        if not ent:
            return None, "no PMC pass recorded for this workload key in profiles/traffic_sidecar.json"
        if world == 1:
            return ent["bytes_per_step"], ent.get("note", "") + f" {src}{stale})"
        if args.scaling == "strong":        # the same set split over N ranks: each GPU moves ~1/N of the 1-GPU pass
            return ent["bytes_per_step"] / world, ent.get("note", "") + f" / {world} ranks, scaled from the 1-GPU PMC pass of the same set {src}{stale})"
        return ent["bytes_per_step"], ent.get("note", "") + f" per rank (every rank fills a shard of the size of the 1-GPU pass) {src}{stale})"
    except Exception as e:
        return None, f"profiles/traffic_sidecar.json unreadable: {e!r}"


def partial_pass(args, local, work):
    from figbird_amd import api, synth
    spec = synth.BenchSpec(mode="partial", read_len=101, insert_mean=180, insert_sd=10, partial_cov=48, gap_mix=args.mix)
    mc = synth.bench_model_case(7, spec)
    mp = synth.write_case(mc, os.path.join(work, "model_partial"))
    model = api.model_from_files(mp["scf"], mp["tmp"], mp["myout"], partial_flag=1, unmapped_flag=0, script_itr=1,
                                 max_distance=spec.max_distance, read_length=spec.read_len, neg_overlap=30, partial_len=mc.partial_len)
    batch, _ = synth.make_bench_batch(args.seed, 8192, spec)      # 8192 gaps: 16 per resident workgroup slot, i.e. steady state
    eng = api.Engine(local)
    eng.set_model(model)
    eng.upload(batch)
    eng.fill_resident()                     # warm-up
    res = eng.fill_resident()
    st = eng.stats()
    eng.free_batch(); eng.close()
    ksec = max(st["kernel_ms"] / 1e3, 1e-9)
    return {"value": batch.n_gaps / ksec, "unit": "gaps/s", "ms_per_step": st["kernel_ms"], "n_gaps": int(batch.n_gaps),
            "reads_per_gap_mean": float(batch.p_read_off[-1]) / max(batch.n_gaps, 1), "filled_bases_per_s": res.filled_bases / ksec,
            "achieved_tflops": st["alg_flops"] / ksec / 1e12, "frac_of_fp64_nofma_peak": st["alg_flops"] / ksec / 1e12 / FP64_NOFMA_PEAK_TFLOPS,
            "workload": "frag-library (2x101 bp, insert 180, 48 soft-clipped reads per gap) partial-mode pass over 8192 gaps of the same gap mix, kernel time of one fill"}


def _run_ref_sample(exe, kind, sample, batch, mc, spec, root, cores, timeout_s):
    """Fill `sample` (gap ids of `batch`) with one CPU process per core; returns (wall, shards, paths, order) or None on timeout."""
    from figbird_amd import synth
    paths = synth.write_batch_subset(batch, sample, mc, root, spec)
    order = paths["gap_order"]
    sset = set(sample)
    sel = [i for i, g in enumerate(order) if g in sset]
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
    ok = True
    for p in procs:
        try:
            p.wait(timeout=max(1.0, timeout_s - (time.perf_counter() - t0)))
        except subprocess.TimeoutExpired:
            ok = False
            p.kill(); p.wait()
    wall = time.perf_counter() - t0
    return (wall, shards, paths, order) if ok else None


def cpu_baseline(args, spec, batch, mc, res, eng, work, st_full, budget_s):
    """The reference (oracle/_ref/Figbird.out, -O2 and as-shipped -O0) or the oracle port on the host cores, on a bounded
    sample of the SAME batch.  A <=400-bp gap of this set costs the reference 10^2-10^3 CPU-seconds, so the sample is
    the >400-bp bracket (one candidate length, a few EM iterations: ~1-2 s per gap per core); the same-mix figure in
    `value` is derived in main() from the reference's measured flop rate and the step's algorithmic flops."""
    from figbird_amd import synth
    from tools import build_test_infra as fbuild    # checker binaries only (oracle / oracle/_ref): the CPU baseline leg
    cores = min(os.cpu_count() or 1, 16)
    k = args.cpu_sample_gaps or cores * 5
    G = np.asarray(batch.gap_len)
    if spec.mode == "unmapped":
        nread = np.diff(batch.u_read_off)
        cand = [int(g) for g in np.argsort(nread, kind="stable") if G[g] > 400]
        label = ">400-bp bracket"
        sample = cand[:k]
    else:
        label = "all brackets"
        sample = list(range(batch.n_gaps))[:max(k * 64, 256)]
    if not sample:
        raise RuntimeError("no gap fits the CPU sample")
    ref = os.path.join(fbuild.REFDIR, "Figbird.out")
    ref0 = os.path.join(fbuild.REFDIR, "Figbird_O0.out")
    kind = "reference" if os.path.exists(ref) else "port"
    exe = [ref] if kind == "reference" else [fbuild.ORACLE, "figbird"]
    r = _run_ref_sample(exe, kind, sample, batch, mc, spec, os.path.join(work, "cpu"), cores, min(budget_s * 0.6, 90.0))
    if r is None:
        raise RuntimeError("CPU sample did not finish inside its budget")
    wall, shards, paths, order = r
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
    nmean = int(np.diff(batch.u_read_off)[sample].mean()) if spec.mode == "unmapped" else int(np.diff(batch.p_read_off)[sample].mean())
    out = {"value": len(sample) / wall, "unit": "gaps/s", "cores": len(shards), "kind": kind,
           "sample": f"{len(sample)} gaps of the batch from the {label} ({int(G[sample].min())}-{int(G[sample].max())} bp, {nmean} reads/gap), "
                     f"one {'oracle/_ref/Figbird.out (-O2 build of the reference)' if kind == 'reference' else 'oracle port'} process per core, {wall:.1f} s wall",
           "measured_sample_gaps_per_s": len(sample) / wall,
           "gflops": st["alg_flops"] / wall / 1e9, "gpu_same_sample_gaps_per_s": len(sample) / max(st["kernel_ms"] / 1e3, 1e-9),
           "gpu_same_sample_gflops": st["alg_flops"] / max(st["kernel_ms"] / 1e3, 1e-9) / 1e9, "parity_on_sample": bool(ok)}
    # as-shipped build (RunFigbird.sh never passes -O): one gap per core, taken evenly across the same sample, compared
    # through flop rates (the sample is sorted by read count, so its gaps differ in cost)
    if kind == "reference" and os.path.exists(ref0) and budget_s - wall > 40:
        s0 = sample[::max(1, len(sample) // cores)][:cores]
        r0 = _run_ref_sample([ref0], kind, s0, batch, mc, spec, os.path.join(work, "cpu_O0"), cores, min(45.0, budget_s - wall - 10))
        if r0 is not None:
            eng.free_batch()
            eng.upload(synth.subset_batch(batch, s0))
            eng.fill_resident()
            st0 = eng.stats()
            eng.free_batch()
            eng.upload(batch)
            g0 = st0["alg_flops"] / r0[0] / 1e9
            out["as_shipped_O0"] = {"gflops": g0, "gaps_per_s": len(s0) / r0[0], "wall_s": r0[0], "slowdown_vs_O2": out["gflops"] / max(g0, 1e-12),
                                    "sample": f"{len(s0)} gaps spread evenly over the same sample, one per core, oracle/_ref/Figbird_O0.out (g++ without -O, as RunFigbird.sh builds it)"}
    return out


if __name__ == "__main__":
    try:
        main()
    finally:
        emit()
