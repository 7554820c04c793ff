"""Multi-GPU sharding of the gap-fill path: gaps are independent (the reference already runs them in separate
processes, FillGaps.cpp:668-679), so ranks take disjoint shards, fill them with no data-path collective, and
ONE all-gather (RCCL over xGMI with backend "nccl", gloo on CPU) reassembles the per-gap results on every rank
before rank 0 rebuilds the scaffold."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def partition_lpt(costs: Sequence[float], world: int) -> List[List[int]]:
    """Longest-processing-time-first deal of gap ids into `world` bins (cost-balanced shards).  The reference's
    own dispatcher separates <=400-bp "critical" gaps from larger ones for the same reason (FillGaps.cpp:523-530)."""
    order = np.argsort(-np.asarray(costs, dtype=np.float64), kind="stable")
    load = np.zeros(world)
    bins: List[List[int]] = [[] for _ in range(world)]
    for g in order:
        r = int(np.argmin(load))
        bins[r].append(int(g))
        load[r] += float(costs[g])
    for b in bins:
        b.sort()
    return bins


def estimate_cost(gap_len: np.ndarray, n_reads: np.ndarray, read_len: int, unmapped: bool, partial_len: int, unm_limit: int = 400) -> np.ndarray:
    """R * W * L * candidates * iterations, with the candidate range of findFrac (Figbird.cpp:6879-6906)."""
    G = np.asarray(gap_len, dtype=np.float64)
    R = np.asarray(n_reads, dtype=np.float64)
    if unmapped:
        cand = np.where(G <= unm_limit // 3, 3.0 * partial_len - 0.3 * G, np.where(G <= unm_limit, 2.0 * G, 1.0))
        # placeReads calls per candidate and the typical candidate length of the bracket, calibrated on the bench batch
        # (tools/cost_model_check.py: profiles/round2/cost_model_check_before.json -> profiles/round3/cost_model_check_after.json)
        its = np.where(G <= unm_limit // 3, 10.5, np.where(G <= unm_limit, 11.5, 2.0))
        Gc = np.where(G <= unm_limit // 3, 0.5 * (0.3 * G + 3.0 * partial_len), np.where(G <= unm_limit, 1.5 * G, G))
        W = np.minimum(Gc + read_len, 2200.0)
    else:
        cand = np.where(G <= partial_len, 3.0 * partial_len, np.where(G <= 2 * partial_len, 5.0 * G, 1.0))
        its = 3.0
        W = float(read_len)
    return R * W * read_len * np.maximum(cand, 1.0) * its + 1.0


def all_status_max(code: int, device=None) -> int:
    """MAX over ranks of a small status code (0 = fine): every rank learns that some rank failed BEFORE the payload
    collective, so that all of them leave together instead of one returning early and the others blocking in all_gather."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return int(code)
    t = torch.tensor([int(code)], dtype=torch.int32, device=device if device is not None else torch.device("cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def all_reduce_or_bits(bits: np.ndarray, device=None) -> np.ndarray:
    """OR over ranks of a small per-gap bit array (each rank sets the bits of its own shard): the one exchange BEFORE the
    fill -- which gaps get to Figbird.cpp:6317 (fig_batch_probe_reach), needed because the gap that sets the reference's
    process-global overlap_threshold may sit on another rank than the gap that reads it."""
    import torch
    import torch.distributed as dist
    a = np.ascontiguousarray(bits, dtype=np.int32)
    if not dist.is_initialized() or dist.get_world_size() == 1 or len(a) == 0:
        return (a != 0).astype(np.uint8)
    t = torch.from_numpy(a.copy())
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return (t.cpu().numpy() != 0).astype(np.uint8)


def all_gather_packed(ids: Sequence[int], res, n_total: int, device=None, extras: Sequence[np.ndarray] = (), force_collective: bool = False):
    """All-gather the shard results as packed byte buffers: one small header exchange, then ONE all-gather of a
    single uint8 buffer per rank laid out [ids | filled_len | gaptofill | extras... | gap strings], straight from the
    numpy arrays the C ABI filled (no per-gap Python strings on the way).  `res` is an api.FillResult (fields
    filled_len, gaptofill, str_off, raw); `extras` are optional int32 arrays of any length that travel with the
    shard (figfill_mp sends the draw planes this way).  Returns (filled_len[n_total], gaptofill[n_total], strings)
    -- `strings` a PackedStrings in global gap order -- and, when extras were given, a fourth element: per rank,
    the tuple (ids, extras...) as received.  A single rank skips the collective unless `force_collective` (a one-rank RCCL
    run of the device-tensor path: tests/test_gpu_parity.py)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_initialized() else 1
    ids_a = np.ascontiguousarray(ids, dtype=np.int32)
    n = len(ids_a)
    fl_a = np.ascontiguousarray(res.filled_len[:n], dtype=np.int32)
    gt_a = np.ascontiguousarray(res.gaptofill[:n], dtype=np.int32)
    nbytes = int(res.str_off[n]) if n else 0
    payload = np.ascontiguousarray(res.raw[:nbytes], dtype=np.uint8)
    ex = [np.ascontiguousarray(e, dtype=np.int32) for e in extras]
    ne = len(ex)
    if world == 1 and not (force_collective and dist.is_initialized()):
        blobs = [(ids_a, fl_a, gt_a, payload, ex)]
    else:
        dev = device if device is not None else torch.device("cpu")
        head = torch.tensor([n, nbytes] + [len(e) for e in ex], dtype=torch.int64, device=dev)
        hs = [torch.zeros(2 + ne, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(hs, head)
        heads = [[int(v) for v in h.cpu()] for h in hs]
        mx = [max(h[k] for h in heads) for k in range(2 + ne)]
        seg = [4 * mx[0]] * 3 + [4 * mx[2 + k] for k in range(ne)] + [mx[1] + 8]
        off = np.concatenate([[0], np.cumsum(seg)]).astype(np.int64)
        mine = np.zeros(int(off[-1]), dtype=np.uint8)
        for k, arr in enumerate([ids_a, fl_a, gt_a] + ex):
            mine[off[k]:off[k] + 4 * len(arr)] = arr.view(np.uint8)
        mine[off[3 + ne]:off[3 + ne] + nbytes] = payload
        buf = torch.from_numpy(mine).to(dev)
        bufs = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(bufs, buf)
        blobs = []
        for r in range(world):
            a = bufs[r].cpu().numpy()
            nr, nb = heads[r][0], heads[r][1]
            i32 = lambda k, cnt: a[off[k]:off[k] + 4 * cnt].view(np.int32)
            blobs.append((i32(0, nr), i32(1, nr), i32(2, nr), a[off[3 + ne]:off[3 + ne] + nb], [i32(3 + k, heads[r][2 + k]) for k in range(ne)]))
    gid = np.concatenate([b[0] for b in blobs]) if blobs else np.zeros(0, dtype=np.int32)
    flc = np.concatenate([b[1] for b in blobs]); gtc = np.concatenate([b[2] for b in blobs])
    pay = np.concatenate([b[3] for b in blobs])
    lens = np.maximum(flc, 0).astype(np.int64)
    src_start = np.cumsum(lens) - lens
    fl = np.zeros(n_total, dtype=np.int32); gt = np.zeros(n_total, dtype=np.int32)
    fl[gid] = flc; gt[gid] = gtc
    glens = np.zeros(n_total, dtype=np.int64); glens[gid] = lens
    off = np.zeros(n_total + 1, dtype=np.int64); off[1:] = np.cumsum(glens)
    perm = np.argsort(gid, kind="stable")
    total = int(off[-1])
    idx = np.repeat(src_start[perm] - off[:-1][gid[perm]], lens[perm]) + np.arange(total, dtype=np.int64)
    raw = pay[idx] if total else np.zeros(0, dtype=np.uint8)
    if ne:
        return fl, gt, PackedStrings(off, raw), [(b[0],) + tuple(b[4]) for b in blobs]
    return fl, gt, PackedStrings(off, raw)


class PackedStrings:
    """Gap strings of a whole gap set in one byte buffer (global gap order)."""

    def __init__(self, off: np.ndarray, raw: np.ndarray):
        self.off, self.raw = off, raw

    def __len__(self):
        return len(self.off) - 1

    def __getitem__(self, g: int) -> bytes:
        return self.raw[self.off[g]:self.off[g + 1]].tobytes()

    def __iter__(self):
        return (self[g] for g in range(len(self)))

    def filled_bases(self) -> int:
        return int(len(self.raw) - np.count_nonzero(self.raw == ord("N")))

    def to_list(self) -> List[str]:
        return [self[g].decode() for g in range(len(self))]


def all_gather_results(ids: Sequence[int], filled_len: np.ndarray, gaptofill: np.ndarray, strings: Sequence[str], n_total: int,
                       device=None) -> Tuple[np.ndarray, np.ndarray, List[str]]:
    """String-list face of all_gather_packed (kept for callers that hold Python strings)."""
    from types import SimpleNamespace
    raw = np.frombuffer("".join(strings).encode() or b"\0", dtype=np.uint8)
    off = np.zeros(len(strings) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(s) for s in strings])
    res = SimpleNamespace(filled_len=np.asarray(filled_len, dtype=np.int32), gaptofill=np.asarray(gaptofill, dtype=np.int32), str_off=off, raw=raw)
    fl, gt, ps = all_gather_packed(ids, res, n_total, device=device)
    return fl, gt, ps.to_list()
