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
        its = np.where(G <= unm_limit, 14.0, 6.0)
        W = np.minimum(G * np.where(G <= unm_limit, 1.5, 1.0) + read_len, 2200.0)
    else:
        cand = np.where(G <= partial_len, 3.0 * partial_len, np.where(G <= 2 * partial_len, 5.0 * G, 1.0))
        its = 3.0
        W = float(read_len)
    return R * W * read_len * np.maximum(cand, 1.0) * its + 1.0


def all_gather_results(ids: Sequence[int], filled_len: np.ndarray, gaptofill: np.ndarray, strings: Sequence[str], n_total: int,
                       device=None) -> Tuple[np.ndarray, np.ndarray, List[str]]:
    """All-gather the shard results: one fixed-size header exchange (counts, byte totals) and one padded payload
    all-gather.  Returns (filled_len[n_total], gaptofill[n_total], strings[n_total]) on every rank."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        fl = np.zeros(n_total, dtype=np.int32); gt = np.zeros(n_total, dtype=np.int32); ss = [""] * n_total
        for k, g in enumerate(ids):
            fl[g] = filled_len[k]; gt[g] = gaptofill[k]; ss[g] = strings[k]
        return fl, gt, ss
    dev = device if device is not None else torch.device("cpu")
    n = len(ids)
    payload = "".join(strings).encode()
    head = torch.tensor([n, len(payload)], dtype=torch.int64, device=dev)
    heads = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(heads, head)
    max_n = int(max(int(h[0]) for h in heads)); max_b = int(max(int(h[1]) for h in heads))
    rec = torch.zeros(max_n * 3 + 1, dtype=torch.int32, device=dev)          # ids, filled_len, gaptofill
    if n:
        rec[:n] = torch.as_tensor(np.asarray(ids, dtype=np.int32), device=dev)
        rec[max_n:max_n + n] = torch.as_tensor(np.asarray(filled_len, dtype=np.int32), device=dev)
        rec[2 * max_n:2 * max_n + n] = torch.as_tensor(np.asarray(gaptofill, dtype=np.int32), device=dev)
    buf = torch.zeros(max(max_b, 1), dtype=torch.uint8, device=dev)
    if payload:
        buf[:len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(dev)
    recs = [torch.zeros_like(rec) for _ in range(world)]
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(recs, rec)
    dist.all_gather(bufs, buf)
    fl = np.zeros(n_total, dtype=np.int32); gt = np.zeros(n_total, dtype=np.int32); ss = [""] * n_total
    for r in range(world):
        nr = int(heads[r][0])
        rc = recs[r].cpu().numpy(); raw = bufs[r].cpu().numpy().tobytes()
        o = 0
        for k in range(nr):
            g = int(rc[k]); L = int(rc[max_n + k])
            fl[g] = L; gt[g] = int(rc[2 * max_n + k])
            ss[g] = raw[o:o + max(L, 0)].decode(); o += max(L, 0)
    return fl, gt, ss
