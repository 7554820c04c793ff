"""N>1 path on CPU: world_size-2 gloo, gaps sharded by the LPT partition, each rank fills its shard through
the C ABI (here backed by the test-only emulation library), one all-gather reassembles the results."""
import os
import socket
import sys

import numpy as np
import pytest

import util
from figbird_amd import dist as fdist


def test_partition_lpt_is_a_balanced_partition():
    rng = np.random.default_rng(3)
    costs = rng.lognormal(3, 2, size=500)
    bins = fdist.partition_lpt(costs, 8)
    flat = sorted(g for b in bins for g in b)
    assert flat == list(range(500))
    loads = [sum(costs[g] for g in b) for b in bins]
    assert max(loads) <= min(loads) + costs.max() + 1e-9


def _worker(rank, world, port, root, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, util.ROOT)
    import torch.distributed as dist
    from figbird_amd import api, synth, build as fbuild
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        case = synth.make_case("partial_brackets", 21, "partial", [(2000, 3), (3500, 50), (5000, 75), (6500, 101), (8200, 260)], read_len=50,
                               insert_mean=180, insert_sd=10, coverage=12, err=0.005, n_model_pairs=800, contig_len=11000)
        full = synth.case_to_batch(case)
        n = full.n_gaps
        nreads = np.diff(full.p_read_off)
        costs = fdist.estimate_cost(full.gap_len, nreads, 50, False, 50)
        mine = fdist.partition_lpt(costs, world)[rank]
        sub = synth.subset_batch(full, mine)
        model = api.model_from_files(os.path.join(root, "scf.fa"), os.path.join(root, "tmp") + "/", os.path.join(root, "tmp", "myout.sam"),
                                     partial_flag=1, unmapped_flag=0, script_itr=1, max_distance=180, read_length=50, neg_overlap=30, partial_len=50)
        eng = api.Engine(0, lib_path=util.EMULIB)
        eng.set_model(model)
        res = eng.fill(sub)
        eng.close()
        fl, gt, ss = fdist.all_gather_results(mine, res.filled_len, res.gaptofill, res.strings, n)
        q.put((rank, mine, fl.tolist(), gt.tolist(), ss))
    finally:
        dist.destroy_process_group()


def test_two_rank_shard_and_gather_reproduces_the_reference(tmp_path):
    import torch.multiprocessing as mp
    root = util.extract_golden("partial_brackets", str(tmp_path))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, 2, port, root, q)) for r in range(2)]
    for p in ps:
        p.start()
    outs = [q.get(timeout=300) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp = [ln.split("\t") for ln in util.read(os.path.join(root, "ref", "gapout.txt")).splitlines()]
    shards = sorted(o[1] for o in outs)
    assert sorted(g for s_ in shards for g in s_) == list(range(len(exp))) and all(len(s_) > 0 for s_ in shards)
    for rank, mine, fl, gt, ss in outs:                     # every rank holds the full, identical result
        assert fl == [int(e[4]) for e in exp]
        assert ss == [e[5] if len(e) > 5 else "" for e in exp]


def _mp_worker(rank, world, port, root, name, q, lib=util.EMULIB):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    sys.path.insert(0, util.ROOT)
    from figbird_amd import figfill_mp
    os.chdir(root)
    rc = figfill_mp.run(util.meta(root)["fillgaps_argv"], backend="gloo", lib_path=lib, device_index=0, verbose=False)
    q.put((rank, rc))


@pytest.mark.parametrize("name,world", [("unmapped_mid_err", 2), ("partial_brackets", 2), ("edge_contig_ends", 3), ("threads3", 2), ("ot_carry", 2), ("ot_carry_t3", 2)])
def test_figfill_mp_writes_the_reference_files(name, world, tmp_path):
    """The multi-GPU product path (figbird_amd.figfill_mp: run handle -> LPT shards -> C ABI -> one packed all-gather ->
    rank 0 writes) on `world` gloo ranks: gapout.txt, draw.txt, filledContigs.fa and Ncount.txt byte-identical to the
    reference's (numthreads=1 goldens)."""
    import torch.multiprocessing as mp
    root = util.extract_golden(name, str(tmp_path))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_mp_worker, args=(r, world, port, root, name, q)) for r in range(world)]
    for p in ps:
        p.start()
    outs = [q.get(timeout=600) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(rc == 0 for _, rc in outs)
    for fn in util.ref_files(root):
        assert util.read(os.path.join(root, "tmp", fn)) == util.read(os.path.join(root, "ref", fn)), fn


def test_figfill_mp_single_rank_without_a_launcher(tmp_path, monkeypatch):
    """N = 1: no process group, same files."""
    from figbird_amd import figfill_mp
    root = util.extract_golden("partial_small", str(tmp_path))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.chdir(root)
    assert figfill_mp.run(util.meta(root)["fillgaps_argv"], lib_path=util.EMULIB, device_index=0, verbose=False) == 0
    for fn in ("gapout.txt", "filledContigs.fa", "Ncount.txt", "draw.txt"):
        assert util.read(os.path.join(root, "tmp", fn)) == util.read(os.path.join(root, "ref", fn)), fn


@pytest.mark.gpu
def test_figfill_mp_on_the_device_single_rank(tmp_path, monkeypatch):
    """The launcher's N = 1 path on the MI355X (libfighip.so through fig_fill_gaps): the reference's four files."""
    from figbird_amd import figfill_mp
    root = util.extract_golden("unmapped_small", str(tmp_path))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.chdir(root)
    assert figfill_mp.run(util.meta(root)["fillgaps_argv"], device_index=0, verbose=False) == 0
    assert "libfighip.so" in open("/proc/self/maps").read()
    for fn in ("gapout.txt", "filledContigs.fa", "Ncount.txt", "draw.txt"):
        assert util.read(os.path.join(root, "tmp", fn)) == util.read(os.path.join(root, "ref", fn)), fn


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["threads3", "ot_carry_t3"])
def test_figfill_mp_two_ranks_on_the_device(name, tmp_path):
    """The launcher's N = 2 path with the real library: two rank processes share the box's one MI355X (gloo for the
    exchanges, since RCCL refuses two ranks on one device), each fills its shard through libfighip.so, rank 0 writes.  On the
    `threads3` / `ot_carry_t3` fixtures, whose per-process overlap_threshold carry crosses the shards (the reach bits measured
    by fig_probe_kernel on one rank decide the preset of a gap on the other)."""
    import torch.multiprocessing as mp
    root = util.extract_golden(name, str(tmp_path))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_mp_worker, args=(r, 2, port, root, name, q, None)) for r in range(2)]
    for p in ps:
        p.start()
    outs = [q.get(timeout=600) for _ in ps]
    for p in ps:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(rc == 0 for _, rc in outs)
    for fn in util.ref_files(root):
        assert util.read(os.path.join(root, "tmp", fn)) == util.read(os.path.join(root, "ref", fn)), fn


def _fail_worker(rank, world, port, root, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    sys.path.insert(0, util.ROOT)
    from figbird_amd import figfill_mp
    os.chdir(root)
    argv = list(util.meta(root)["fillgaps_argv"])
    if rank == 1:
        argv[8] = "no_such_tmp/"              # this rank cannot open its inputs
    q.put((rank, figfill_mp.run(argv, backend="gloo", lib_path=util.EMULIB, device_index=0, verbose=False)))


def test_figfill_mp_ranks_fail_together(tmp_path):
    """A rank that cannot open its run must not leave its peer blocked in the all-gather: the status all-reduce in front
    of the payload collective makes both return non-zero (the reference ignores its workers' exit codes,
    FillGaps.cpp:133; the launcher must not hang instead)."""
    import torch.multiprocessing as mp
    root = util.extract_golden("partial_small", str(tmp_path))
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_fail_worker, args=(r, 2, port, root, q)) for r in range(2)]
    for p in ps:
        p.start()
    outs = dict(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(timeout=60)
    assert outs == {0: 1, 1: 1}


def test_all_gather_packed_forced_collective_single_rank_gloo():
    """The collective path of all_gather_packed with ONE rank (no world == 1 short-cut): header exchange, padded buffer,
    all_gather, unpack -- the code an N-GPU run executes, here on gloo."""
    import torch.distributed as dist
    from types import SimpleNamespace
    from figbird_amd import dist as fdist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    try:
        strings = ["ACGT", "", "NNACG", "T"]
        raw = np.frombuffer("".join(strings).encode(), dtype=np.uint8)
        off = np.zeros(5, dtype=np.int64); off[1:] = np.cumsum([len(x) for x in strings])
        res = SimpleNamespace(filled_len=np.array([4, 0, 5, 1], dtype=np.int32), gaptofill=np.array([0, 7, 0, 0], dtype=np.int32), str_off=off, raw=raw)
        fl, gt, ps, per = fdist.all_gather_packed([2, 0, 3, 1], res, 4, extras=[np.arange(6, dtype=np.int32)], force_collective=True)
        assert list(fl) == [0, 1, 4, 5] and list(gt) == [7, 0, 0, 0]
        assert ps.to_list() == ["", "T", "ACGT", "NNACG"]
        assert list(per[0][1]) == list(range(6))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_all_gather_packed_single_rank_on_the_device():
    """RCCL itself: backend "nccl" with one rank on cuda:0, device tensors, the collective path forced -- communicator
    init, all_reduce (the status word), all_gather of the header and of the packed buffer run once on the MI355X before an
    8-GPU node sees them."""
    import subprocess
    code = r'''
import os, sys, socket
import numpy as np
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from types import SimpleNamespace
from figbird_amd import dist as fdist
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == "nccl"
assert fdist.all_status_max(0, dev) == 0
t = torch.ones(1, device=dev); dist.all_reduce(t); assert int(t.item()) == 1
strings = ["ACGT" * 300, "", "NNACG", "T"]
raw = np.frombuffer("".join(strings).encode(), dtype=np.uint8)
off = np.zeros(5, dtype=np.int64); off[1:] = np.cumsum([len(x) for x in strings])
res = SimpleNamespace(filled_len=np.array([1200, 0, 5, 1], dtype=np.int32), gaptofill=np.array([0, 7, 0, 0], dtype=np.int32), str_off=off, raw=raw)
fl, gt, ps, per = fdist.all_gather_packed([2, 0, 3, 1], res, 4, device=dev, extras=[np.arange(6, dtype=np.int32)], force_collective=True)
assert list(fl) == [0, 1, 1200, 5] and list(gt) == [7, 0, 0, 0], (fl, gt)
assert ps.to_list() == ["", "T", "ACGT" * 300, "NNACG"]
assert list(per[0][1]) == list(range(6))
dist.barrier()
dist.destroy_process_group()
print("RCCL_OK")
''' % util.ROOT
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stderr[-3000:]


@pytest.mark.gpu
def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it starts two rank processes itself; on the
    one-GPU box both ranks share the card (FIGBENCH_DEVICE=0) and the collectives run on gloo.  The line must report
    n_gpus = 2 and both ranks' kernel times."""
    import json, subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(FIGBENCH_BACKEND="gloo", FIGBENCH_DEVICE="0")
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--gaps-per-gpu", "24",
                        "--reads-per-gap", "60", "--cpu-baseline", "0", "--partial-pass", "0"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1, r.stdout
    d = json.loads(line[0])
    # default scaling is strong: ONE fixed set of 8 x gaps-per-gpu gaps whatever N is, split over the ranks
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["n_gaps"] == 8 * 24 and len(d["roofline"]["per_rank_kernel_ms_per_step"]) == 2 and d["value"] > 0


def _cfg4_worker(rank, world, port, workdir, q):
    """One rank of the config-4 rehearsal: same seeded gap set on every rank, LPT shard, fill through the C ABI on the box's
    one GPU, gloo all-gather of the packed results; rank 0 checks them."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    sys.path.insert(0, util.ROOT)
    import subprocess
    import torch.distributed as dist
    from figbird_amd import api, synth
    dist.init_process_group("gloo")
    try:
        n_gaps, L = 2048, 100
        spec = synth.BenchSpec(mode="unmapped", read_len=L, insert_mean=2500.0, insert_sd=250.0, reads_per_gap_mean=48.0, frag_len=L)
        mc = synth.bench_model_case(7, spec)
        mp = synth.write_case(mc, os.path.join(workdir, f"model_r{rank}"))
        model = api.model_from_files(mp["scf"], mp["tmp"], mp["myout"], partial_flag=0, unmapped_flag=1, script_itr=1,
                                     max_distance=spec.max_distance, read_length=L, neg_overlap=30, partial_len=mc.partial_len)
        batch, truth = synth.make_bench_batch(4004, n_gaps, spec)
        cost = fdist.estimate_cost(np.asarray(batch.gap_len), np.diff(batch.u_read_off), L, True, mc.partial_len)
        shards = fdist.partition_lpt(cost, world)
        mine = shards[rank]
        eng = api.Engine(0)
        eng.set_model(model)
        res = eng.fill(synth.subset_batch(batch, mine))
        eng.close()
        fl, gt, ps = fdist.all_gather_packed(mine, res, n_gaps)
        out = {"rank": rank, "n_mine": len(mine), "native": "libfighip.so" in open("/proc/self/maps").read()}
        if rank == 0:
            strings = ps.to_list()
            called = wrong = 0
            for s_, t in zip(strings, truth):
                if len(s_) == len(t):
                    a = np.frombuffer(s_.encode(), dtype=np.uint8)
                    m = a != ord("N")
                    called += int(m.sum()); wrong += int((a[m] != t[m]).sum())
            out.update(called=called, wrong=wrong, n_strings=len(strings), filled=int(ps.filled_bases()))
            # stratified oracle sample: the cheapest gap of each findFrac bracket (one of them from each rank's shard at least)
            G = np.asarray(batch.gap_len); nr = np.diff(batch.u_read_off)
            sample = []
            for lo, hi in [(5, 31), (31, 134), (134, 401), (401, 800), (800, 2001)]:
                ids = [g for g in range(n_gaps) if lo <= G[g] < hi]
                sample.append(min(ids, key=lambda g: (int(nr[g]) * (int(G[g]) if G[g] <= 400 else 1), g)))
            paths = synth.write_batch_subset(batch, sample, mc, os.path.join(workdir, "cpu"), spec)
            args = [paths["scf"], str(spec.max_distance), str(L), "1", "0", "1", "1", paths["myout"], paths["tmp"], paths["gaps"],
                    "30", str(mc.partial_len), "10", "0", str(int(spec.insert_mean))]
            r = subprocess.run([util.ORACLE, "fillgaps"] + args, cwd=workdir, capture_output=True, text=True, timeout=900)
            ok = r.returncode == 0
            lines = open(paths["tmp"] + "gapout.txt").read().splitlines() if ok else []
            for k, g in enumerate(paths["gap_order"]):
                if ok and g in sample:
                    f = lines[k].split("\t")
                    ok = int(f[4]) == int(fl[g]) and (f[5] if len(f) > 5 else "") == strings[g]
            out.update(oracle_identical=bool(ok), sample=[int(g) for g in sample], sample_owner=[int(g in set(mine)) for g in sample])
        dist.barrier()
        q.put(out)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_config4_shape_two_ranks_shard_fill_gather(tmp_path):
    """BASELINE config 4's shape (chr14-like: thousands of gaps, 2x100-bp jump reads) through the multi-rank path: 2 048 gaps of
    the GAGE mix dealt LPT on estimated cost to two ranks (the role of FillGaps.cpp:456-649), each rank fills its shard on the
    device, one packed all-gather reassembles them; every called base equals the synthetic truth (to the error rate), and a
    stratified sample (one gap per findFrac bracket) equals the oracle byte for byte."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_cfg4_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in ps:
        p.start()
    outs = {o["rank"]: o for o in (q.get(timeout=1500) for _ in ps)}
    for p in ps:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert outs[0]["native"] and outs[1]["native"]
    assert outs[0]["n_mine"] + outs[1]["n_mine"] == 2048 and min(outs[0]["n_mine"], outs[1]["n_mine"]) > 600
    r0 = outs[0]
    assert r0["n_strings"] == 2048 and r0["called"] > 100000 and r0["wrong"] <= 0.003 * r0["called"], r0
    assert r0["oracle_identical"], r0
    assert 0 < sum(r0["sample_owner"]) < len(r0["sample_owner"])          # the sample spans both ranks' shards


def _figfill_devices(root, exe, devices, serial=False):
    env = {"FIGFILL_DEVICES": devices}
    if serial:
        env["FIGFILL_SERIAL"] = "1"          # the one-lane emulation library is not re-entrant; libfighip.so's contexts are
    return util.run([exe] + util.meta(root)["fillgaps_argv"], root, env)


@pytest.mark.parametrize("name,devices", [("threads3", "0,0"), ("unmapped_mid_err", "0,0,0"), ("partial_brackets", "0,0"), ("ot_carry", "0,0"), ("ot_carry_t3", "0,0,0")])
def test_figfill_devices_from_the_cpp_host_emulation(name, devices, tmp_path):
    """N GPUs from the C++ host (FIGFILL_DEVICES, figfill_main.cpp: fill_multi): LPT shards in C++, one fig_ctx and host thread
    per listed device, results merged in host memory -- here on the one-lane emulation library (every "device" is ordinal 0),
    byte-identical to the reference's four files (and gaploads.txt for `threads3`, whose per-process overlap_threshold presets
    travel with the shards)."""
    root = util.extract_golden(name, str(tmp_path))
    r = _figfill_devices(root, util.EMU, devices, serial=True)
    assert r.returncode == 0, r.stderr
    assert f"{len(devices.split(','))} GPUs" in r.stdout
    for fn in util.ref_files(root):
        assert util.read(os.path.join(root, "tmp", fn)) == util.read(os.path.join(root, "ref", fn)), fn


def test_cpp_partition_matches_the_python_partition(tmp_path):
    """fighost::estimate_cost / partition_lpt (C++) deal the gaps exactly as figbird_amd/dist.py does: figfill with
    FIGFILL_DEVICES and figfill_mp put the same gaps on the same rank."""
    import ctypes as C
    from figbird_amd import api
    root = util.extract_golden("threads3", str(tmp_path))
    h = api.load_host_library()
    h.fighost_partition.argtypes = [C.POINTER(C.c_char_p), C.c_int, api.c_i32_p]
    h.fighost_partition.restype = C.c_int64
    argv = util.meta(root)["fillgaps_argv"]
    owner = np.full(64, -1, dtype=np.int32)
    cwd = os.getcwd(); os.chdir(root)
    try:
        n = h.fighost_partition((C.c_char_p * 15)(*[a.encode() for a in argv]), 3, api._p(owner, api.c_i32_p))
        assert n == util.meta(root)["n_gaps"]
        hh = h.fighost_run_open((C.c_char_p * 15)(*[a.encode() for a in argv]), C.create_string_buffer(256), 256)
        glen = np.zeros(n, dtype=np.int32); nu = np.zeros(n, dtype=np.int64); npp = np.zeros(n, dtype=np.int64)
        h.fighost_run_sizes(hh, api._p(glen, api.c_i32_p), api._p(nu, api.c_i64_p), api._p(npp, api.c_i64_p))
        par = np.zeros(6, dtype=np.int32); h.fighost_run_params(hh, api._p(par, api.c_i32_p))
        h.fighost_run_close(hh)
    finally:
        os.chdir(cwd)
    cost = fdist.estimate_cost(glen, npp if not par[2] else nu, int(par[0]), bool(par[2]), int(par[1]), int(par[3]))
    py = fdist.partition_lpt(cost, 3)
    assert [sorted(int(g) for g in range(n) if owner[g] == r) for r in range(3)] == py


@pytest.mark.gpu
def test_figfill_devices_two_contexts_on_the_device(tmp_path):
    """FIGFILL_DEVICES on the MI355X box: two fig_ctx of the real library fill their shards concurrently from two host threads
    (both on the box's one GPU), merged by the C++ host: the reference's files, byte for byte."""
    for k, name in enumerate(("threads3", "ot_carry", "ot_carry_t3")):        # ot_carry*: the reach bits of one shard decide the other's carry
        root = util.extract_golden(name, str(tmp_path / f"g{k}"))
        r = _figfill_devices(root, util.FIGFILL, "0,0")
        assert r.returncode == 0, r.stderr
        for fn in util.ref_files(root):
            assert util.read(os.path.join(root, "tmp", fn)) == util.read(os.path.join(root, "ref", fn)), (name, fn)
    root2 = util.extract_golden("bench_b25", str(tmp_path / "b"))
    r = _figfill_devices(root2, util.FIGFILL, "0,0")
    assert r.returncode == 0, r.stderr
    for fn in util.ref_files(root2):
        assert util.read(os.path.join(root2, "tmp", fn)) == util.read(os.path.join(root2, "ref", fn)), fn
