"""figfill over N GPUs of one node -- the multi-GPU face of the FillGaps.cpp drop-in.

    python -m figbird_amd.figfill_mp --gpus N <the 15 FillGaps arguments>          (starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        -m figbird_amd.figfill_mp <the 15 FillGaps arguments>                       (under a launcher)

One process per GPU.  Every rank opens the run through libfighost (same inputs, same model, built once per rank as the
reference builds it once per thread), the gap set is dealt into N shards by `dist.partition_lpt` on `dist.estimate_cost`
(the role of FillGaps.cpp:456-649: the reference separates the costly <= 400-bp gaps from the rest for the same reason),
each rank fills its shard through the C ABI (libfighip.so) with NO data-path collective -- one small all-reduce before
the fill carries the reference's process-global overlap_threshold across shards (fig_batch_probe_reach) -- and ONE
all-gather of packed byte buffers (RCCL over xGMI with backend "nccl"; gloo on CPU for the tests) hands rank 0 what it
needs to write gapout.txt, draw.txt, filledContigs.fa and Ncount.txt -- byte-identical to the single-GPU figfill.
N = 1 (no launcher needed) degenerates to that.  There is no CPU compute path: the fill is libfighip's."""
from __future__ import annotations

import ctypes as C
import os
import sys
import time

import numpy as np

from . import api, dist as fdist


def run(argv15, backend=None, lib_path=None, device_index=None, verbose=True) -> int:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")      # one hardware queue per scheduler lane; before the HIP runtime starts (fig_ctx_create)
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0")) if device_index is None else device_index
    t0 = time.time()
    own_pg = False
    if world > 1 and not dist.is_initialized():
        be = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if be == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(be)
        own_pg = True
    dev = torch.device("cuda", local) if (world > 1 and dist.get_backend() == "nccl") else None
    host = api.load_host_library()
    args = (C.c_char_p * 15)(*[a.encode() for a in argv15])
    err = C.create_string_buffer(512)
    h = host.fighost_run_open(args, err, 512)
    if not h:
        sys.stderr.write(err.value.decode() + "\n")
    if fdist.all_status_max(0 if h else 1, dev) != 0:        # every rank leaves together (a peer must not wait in the all-gather)
        if h:
            host.fighost_run_close(h)
        if own_pg:
            dist.destroy_process_group()
        return 1
    try:
        # Every stage between two collectives runs under try/except: an exception on one rank becomes a status code that
        # all ranks learn at the next all_status_max, so nobody is left waiting in a collective for a peer that has gone.
        rc = 0
        n = 0; mine = np.zeros(0, dtype=np.int64); eng = None; reach = np.zeros(0, dtype=np.uint8); su = C.c_int64(); sp = C.c_int64(); cb = api.FigGapBatch()
        tk = time.time()
        try:
            n = int(host.fighost_run_ngaps(h))
            glen = np.zeros(max(n, 1), dtype=np.int32); nu = np.zeros(max(n, 1), dtype=np.int64); npp = np.zeros(max(n, 1), dtype=np.int64)
            host.fighost_run_sizes(h, api._p(glen, api.c_i32_p), api._p(nu, api.c_i64_p), api._p(npp, api.c_i64_p))
            par = np.zeros(6, dtype=np.int32); host.fighost_run_params(h, api._p(par, api.c_i32_p))
            L, partial_len, unmapped, unm_limit, nmsg = int(par[0]), int(par[1]), int(par[2]), int(par[3]), int(par[4])
            if rank == 0 and verbose:
                print(f"Total # of gaps = {n}")
                for i in range(nmsg):
                    print(host.fighost_run_message(h, i).decode())
            cost = fdist.estimate_cost(glen[:n], (nu if unmapped else npp)[:n], L, bool(unmapped), partial_len, unm_limit)
            shards = fdist.partition_lpt(cost, world)
            mine = np.asarray(shards[rank], dtype=np.int64)
            cm = api.FigModel(); host.fighost_run_model(h, C.byref(cm))
            if host.fighost_run_shard(h, api._p(mine if len(mine) else np.zeros(1, dtype=np.int64), api.c_i64_p), len(mine), C.byref(cb), C.byref(su), C.byref(sp)) != 0:
                raise RuntimeError("bad shard")
            # upload the shard and measure which of its gaps get to Figbird.cpp:6317
            eng = api.Engine(local, lib_path=lib_path)
            eng.set_model_struct(cm)
            eng.upload_struct(cb)
            reach = np.zeros(max(n, 1), dtype=np.uint8)
            if len(mine):
                reach[mine] = eng.probe_reach()
        except Exception as e:       # e.g. FIG_ENOMEM on one GPU: reported to every rank below
            sys.stderr.write(f"figfill_mp: rank {rank}: {e}\n")
            rc = 1
        if fdist.all_status_max(rc, dev) != 0:
            return 1
        # the one exchange before the fill: the bits of every shard; then the carry along the reference's worker processes
        reach = fdist.all_reduce_or_bits(reach, dev)
        res = st = None
        try:
            preset = np.zeros(max(n, 1), dtype=np.uint8)
            host.fighost_run_ot_presets(h, api._p(reach, api.c_u8_p), api._p(preset, api.c_u8_p))
            eng.set_ot_preset(preset[mine] if len(mine) else np.zeros(0, dtype=np.uint8))
            res = eng.fill_struct(cb, int(su.value), int(sp.value), draw=True, resident=True)
            st = eng.stats()
            eng.close()
        except Exception as e:
            sys.stderr.write(f"figfill_mp: rank {rank}: {e}\n")
            rc = 1
        if fdist.all_status_max(rc, dev) != 0:
            return 1
        # The gather's own collective cannot be guarded (a rank that never enters it leaves the others waiting), but everything
        # this rank does before entering it -- taking the draw planes, packing -- happens inside all_gather_packed before its
        # first collective call, so a rank that fails there is reported through the status exchange below instead of a hang:
        # the ranks agree first that all of them hold a result to pack.
        prc = 0 if (res is not None and res.draw is not None) else 1
        if fdist.all_status_max(prc, dev) != 0:
            return 1
        dpos, disz, dlen = res.draw
        out = fdist.all_gather_packed(mine, res, n, device=dev, extras=[dlen, dpos, disz])
        wrc = 0
        try:
            fl, gt, ps, per_rank = out
            if rank == 0:
                # scatter the per-read draw planes back to global read order: [all unmapped reads..., all partial reads...]
                uo = np.zeros(n + 1, dtype=np.int64); uo[1:] = np.cumsum(nu[:n]); po = np.zeros(n + 1, dtype=np.int64); po[1:] = np.cumsum(npp[:n])
                NU, NP = int(uo[-1]), int(po[-1])
                g_pos = np.full(max(NU + NP, 1), np.iinfo(np.int32).min, dtype=np.int32); g_isz = np.zeros(max(NU + NP, 1), dtype=np.int32)
                g_len = np.full(max(2 * n, 1), -1, dtype=np.int32)
                for ids, r_len, r_pos, r_isz in per_rank:
                    ids = np.asarray(ids, dtype=np.int64)
                    if len(ids) == 0:
                        continue
                    g_len[2 * ids] = r_len[0::2]; g_len[2 * ids + 1] = r_len[1::2]
                    cu = nu[ids]; cp = npp[ids]
                    su_ = int(cu.sum())
                    src_u = np.arange(su_, dtype=np.int64)
                    dst_u = np.repeat(uo[ids] - (np.cumsum(cu) - cu), cu) + src_u
                    g_pos[dst_u] = r_pos[:su_]; g_isz[dst_u] = r_isz[:su_]
                    sp_ = int(cp.sum())
                    src_p = np.arange(sp_, dtype=np.int64)
                    dst_p = NU + np.repeat(po[ids] - (np.cumsum(cp) - cp), cp) + src_p
                    g_pos[dst_p] = r_pos[su_:su_ + sp_]; g_isz[dst_p] = r_isz[su_:su_ + sp_]
                raw = ps.raw if len(ps.raw) else np.zeros(1, dtype=np.uint8)
                wrc = host.fighost_run_write(h, api._p(fl if n else np.zeros(1, np.int32), api.c_i32_p), api._p(gt if n else np.zeros(1, np.int32), api.c_i32_p),
                                             api._p(ps.off, api.c_i64_p), C.cast(raw.ctypes.data, C.c_char_p),
                                             api._p(g_pos, api.c_i32_p), api._p(g_isz, api.c_i32_p), api._p(g_len, api.c_i32_p), err, 512)
                if wrc != 0:
                    sys.stderr.write(err.value.decode() + "\n")
                if wrc == 0 and verbose:
                    print(f"Time taken = {time.time() - t0:g} seconds ({world} rank(s); rank 0: {len(mine)} gaps, device kernels {st['kernel_ms']:.3f} ms, fill {time.time() - tk:.3f} s)")
                    print("======================================")
                    print(f"Iteration {int(argv15[3])} ends successfully")
                    print("======================================")
        except Exception as e:
            sys.stderr.write(f"figfill_mp: rank {rank}: {e}\n")
            wrc = 1
        return 1 if fdist.all_status_max(wrc, dev) != 0 else 0        # doubles as the closing barrier
    finally:
        if eng is not None:
            eng.close()                  # idempotent: also reached on the early returns above
        host.fighost_run_close(h)
        if own_pg:
            dist.destroy_process_group()


def launch_ranks(n: int, argv15) -> int:
    """`--gpus N` without a launcher: run the N ranks as a child `torch.distributed.run` (the reference starts its own
    workers too, FillGaps.cpp:668-679).  This parent never touches the GPU; it returns the child's exit code."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "-m", "figbird_amd.figfill_mp"] + list(argv15)
    return subprocess.call(cmd, env=env)


def main():
    if len(sys.argv) >= 3 and sys.argv[1] == "--gpus":
        n = int(sys.argv[2])
        del sys.argv[1:3]
        if n > 1 and "WORLD_SIZE" not in os.environ:
            if len(sys.argv) < 16:
                sys.stderr.write("figfill_mp: --gpus N needs the 15 FillGaps arguments\n")
                sys.exit(1)
            sys.exit(launch_ranks(n, sys.argv[1:16]))
        if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != n:
            sys.stderr.write(f"figfill_mp: --gpus {n} under a launcher with WORLD_SIZE={os.environ['WORLD_SIZE']}\n")
            sys.exit(1)
    if len(sys.argv) < 16:
        sys.stderr.write("usage: [torchrun ...] -m figbird_amd.figfill_mp <contigs.fa> <maxDistance> <readLen> <scriptItr> <partialFlag> <unmapped> "
                         "<numThreads> <myout.sam> <tmp/> <gaps/> <negOverlap> <partialReadLen> <trim> <setInputMean> <insertSize>\n")
        sys.exit(1)
    # FIGFILL_MP_DEVICE=<n>: every rank on GPU n; FIGFILL_MP_BACKEND=gloo: host-side all-gather (rehearsal on a box with fewer
    # GPUs than ranks -- RCCL refuses two ranks on one device).  figfill's own selector FIGFILL_DEVICE is deliberately NOT read
    # here: left exported, it would pin every rank of a real multi-GPU run to one card.
    dev = os.environ.get("FIGFILL_MP_DEVICE")
    sys.exit(run(sys.argv[1:16], backend=os.environ.get("FIGFILL_MP_BACKEND"), device_index=int(dev) if dev is not None else None))


if __name__ == "__main__":
    main()
