"""Parity proper: the HIP engine on a real MI355X, called through the C ABI (ctypes -> libfighip.so, and the
figfill host binary), against (i) the golden vectors from the reference's own binaries, (ii) the oracle on
fresh seeded inputs, and (iii) at bench sizes, size-independent properties."""
import os

import numpy as np
import pytest

import util
from figbird_amd import api, synth

pytestmark = pytest.mark.gpu


def _loaded_native():
    maps = open("/proc/self/maps").read()
    return "libfighip.so" in maps


@pytest.mark.parametrize("name", util.GOLDEN_CASES + util.BENCH_GOLDENS)
def test_figfill_on_gpu_matches_reference_outputs(name, tmp_path):
    """figfill (shipped host + libfighip.so) on the reference's own golden outputs: byte-identical files."""
    root = util.extract_golden(name, str(tmp_path))
    r = util.run_figfill(root, util.FIGFILL)
    assert r.returncode == 0, r.stderr
    for fn in util.ref_files(root):
        assert util.read(os.path.join(root, "tmp", fn)) == util.read(os.path.join(root, "ref", fn)), fn


@pytest.mark.parametrize("name", ["unmapped_small", "unmapped_mid_err", "bench_b25", "bench_c1100"])
def test_pair_chain_estep_still_matches_reference_outputs(name, tmp_path):
    """FIG_ESTEP=pair: the pair-chain E-step (fig_hot_estep) everywhere -- the form the 1217-1600-column class, candidates over
    875 bp and clipped gaps still take -- on fixtures the default build fills through the shared-factor form: same bytes."""
    root = util.extract_golden(name, str(tmp_path))
    r = util.run([util.FIGFILL] + util.meta(root)["fillgaps_argv"], root, {"FIG_ESTEP": "pair"})
    assert r.returncode == 0, r.stderr
    for fn in util.ref_files(root):
        assert util.read(os.path.join(root, "tmp", fn)) == util.read(os.path.join(root, "ref", fn)), fn


def _model_for(root):
    a = util.meta(root)["fillgaps_argv"]
    return api.model_from_files(os.path.join(root, "scf.fa"), os.path.join(root, "tmp") + "/", os.path.join(root, "tmp", "myout.sam"),
                                partial_flag=int(a[4]), unmapped_flag=int(a[5]), script_itr=int(a[3]), max_distance=int(a[1]),
                                read_length=int(a[2]), neg_overlap=int(a[10]), partial_len=int(a[11]))


@pytest.mark.parametrize("name", ["unmapped_small", "unmapped_mid_err", "partial_brackets", "neg_overlap", "repeat_flanks", "cap_3001", "stat2_hint", "model_indels"])
def test_c_abi_planes_within_tolerance(name, tmp_path):
    """Through the C ABI: filled bases and gaptofill exact; per-candidate likelihood (likelihood_arr,
    Figbird.cpp:6390-6391) within 1e-6 relative of the oracle (device libm vs glibc differ in the last ulp),
    EM iteration counts and valid_count exact."""
    from tools.make_golden import make
    root = util.extract_golden(name, str(tmp_path))
    tr = str(tmp_path / "o.trace")
    assert util.run_oracle_fillgaps(root, trace=tr).returncode == 0
    oc, _ = util.parse_trace(tr)
    case = make(name)
    eng = api.Engine(0)
    assert _loaded_native(), "native libfighip.so must be the code that runs"
    eng.set_model(_model_for(root))
    res = eng.fill(synth.case_to_batch(case), debug_cand=2048)
    eng.close()
    exp = [ln.split("\t") for ln in util.read(os.path.join(root, "ref", "gapout.txt")).splitlines()]
    assert [int(e[4]) for e in exp] == list(res.filled_len)
    assert [e[5] if len(e) > 5 else "" for e in exp] == res.strings
    for g, cands in oc.items():
        got = res.cand[g]
        assert len(got) == len(cands), f"gap {g}: candidate count"
        for (G1, it1, lik1, v1), (G2, it2, v2, lik2) in zip(cands, got):
            assert (G1, it1, v1) == (G2, it2, v2), f"gap {g} G={G1}"
            if np.isfinite(lik1):
                assert abs(lik1 - lik2) <= 1e-6 * max(1.0, abs(lik1)), f"gap {g} G={G1}: {lik1} vs {lik2}"
            else:
                assert lik1 == lik2 or (np.isnan(lik1) and np.isnan(lik2))


@pytest.mark.parametrize("seed", list(range(500, 524)))
def test_gpu_fuzz_against_oracle(seed, tmp_path):
    """Fresh seeded inputs (all findFrac brackets, both modes, errors, N in reads, negative overlaps)."""
    from tools.fuzz_ref import mk
    from tools.compare_emu import run_one
    assert run_one(mk(seed), str(tmp_path), exe=util.FIGFILL, verbose=False)


@pytest.mark.parametrize("name", util.BENCH_GOLDENS)
def test_bench_regime_candidates_and_planes_match_the_pinned_oracle(name, tmp_path, monkeypatch):
    """The regime bench.py's step lives in (L = 150, insert N(3500, 350), 600-3000 reads per gap, 270-320 candidate lengths x
    8-16 EM iterations through the speculative scheduler and its five early-stop rules, Figbird.cpp:6298-6482), against
    fixtures made ONCE with the reference's own binaries (tools/make_bench_golden.py): through the C ABI, every candidate
    record (gapEstimate, EM iterations, valid_count exact; likelihood <= 1e-6) and, for a spread of candidates, planes (i)
    countsGap and (ii) the per-read E-step maxima <= 1e-6 against the trace of the oracle whose text outputs equalled the
    reference's bytes on these very inputs.  bench_c1100 / bench_t1800 cover the one-weight-row and the LDS-tiled class."""
    import json
    from tools import make_bench_golden as mbg
    root = util.extract_golden(name, str(tmp_path))
    ref = json.load(open(os.path.join(root, "ref", "cands.json")))
    planes = np.load(os.path.join(root, "ref", "planes.npz"))
    batch, mc, spec = mbg.make(name)
    a = util.meta(root)["fillgaps_argv"]
    model = api.model_from_files(os.path.join(root, "scf.fa"), os.path.join(root, "tmp") + "/", os.path.join(root, "tmp", "myout.sam"),
                                 partial_flag=int(a[4]), unmapped_flag=int(a[5]), script_itr=int(a[3]), max_distance=int(a[1]),
                                 read_length=int(a[2]), neg_overlap=int(a[10]), partial_len=int(a[11]))
    cols = max(int(c[0]) for c in ref["cands"]) + 1
    nreads = int(batch.u_read_off[1])
    eng = api.Engine(0)
    assert _loaded_native()
    eng.set_model(model)
    res = eng.fill(batch, debug_cand=512, plane_cols=cols, plane_reads=nreads)
    st = eng.stats()
    eng.close()
    exp = [ln.split("\t") for ln in util.read(os.path.join(root, "ref", "gapout.txt")).splitlines()]
    assert [int(e[4]) for e in exp] == list(res.filled_len)
    assert [e[5] if len(e) > 5 else "" for e in exp] == res.strings
    got = res.cand[0]
    assert len(got) == len(ref["cands"]), "candidate count"
    for (G1, it1, likhex, v1), (G2, it2, v2, lik2) in zip(ref["cands"], got):
        lik1 = float.fromhex(likhex)
        assert (G1, it1, v1) == (G2, it2, v2), f"G={G1}"
        if np.isfinite(lik1):
            assert abs(lik1 - lik2) <= 1e-6 * max(1.0, abs(lik1)), f"G={G1}: {lik1} vs {lik2}"
        else:
            assert lik1 == lik2 or (np.isnan(lik1) and np.isnan(lik2))
    from test_planes import _close
    index = {c[0]: k for k, c in enumerate(got)}
    assert len(ref["plane_cands"]) >= 1
    for G in ref["plane_cands"]:
        k = index[G]
        cnt, rmax = planes[f"counts_{G}"], planes[f"rmax_{G}"]
        ok, at = _close(res.counts[0, k, :G, :].reshape(-1), cnt.reshape(-1), 1e-6)
        assert ok, f"G={G}: countsGap column {at // 5} base {at % 5}: {res.counts[0, k, at // 5, at % 5]!r} vs {cnt.reshape(-1)[at]!r}"
        ok, at = _close(res.read_maxlv[0, k, :len(rmax)], rmax, 1e-6)
        assert ok, f"G={G}: read {at}: {res.read_maxlv[0, k, at]!r} vs {rmax[at]!r}"
    if ref.get("stats"):          # the oracle's placeReads calls and algorithmic flops on this gap = the device's useful-work counters
        assert int(ref["stats"][0]) == int(st["place_calls"])
        assert abs(float(ref["stats"][1]) - st["alg_flops"]) <= 1e-9 * float(ref["stats"][1])


@pytest.mark.parametrize("seed", [701, 702, 703])
def test_gpu_fuzz_mid_bracket_against_oracle(seed, tmp_path):
    """Fresh seeded unmapped gaps of the 134-400 bracket (candidates 0.5 G0 .. 2.5 G0: hundreds of candidate lengths through
    the speculative scheduler and its early-stop replay), which the general fuzz generator leaves out for its CPU cost."""
    from tools.fuzz_ref import mk_mid
    from tools.compare_emu import run_one
    assert run_one(mk_mid(seed), str(tmp_path), exe=util.FIGFILL, verbose=False)


def _bench_engine(spec, seed=7):
    import tempfile
    mc = synth.bench_model_case(seed, spec)
    d = tempfile.mkdtemp(prefix="figmodel_")
    p = synth.write_case(mc, d)
    m = api.model_from_files(p["scf"], p["tmp"], p["myout"], partial_flag=1 if spec.mode == "partial" else 0,
                             unmapped_flag=1 if spec.mode == "unmapped" else 0, script_itr=1, max_distance=spec.max_distance,
                             read_length=spec.read_len, neg_overlap=30, partial_len=mc.partial_len)
    eng = api.Engine(0)
    eng.set_model(m)
    return eng, mc


def test_bench_shape_properties_unmapped():
    """BASELINE-shaped unmapped batch (2x150 jump library, GAGE-like gap mix): the fill is deterministic
    (two passes over the resident batch give identical bytes), long gaps keep the reference's conservative N
    core, and with these error rates essentially every called base equals the truth."""
    spec = synth.BenchSpec(mode="unmapped", reads_per_gap_mean=40.0)
    eng, _ = _bench_engine(spec)
    batch, truth = synth.make_bench_batch(123, 192, spec)
    eng.upload(batch)
    r1 = eng.fill_resident()
    r2 = eng.fill_resident()
    eng.free_batch(); eng.close()
    assert r1.strings == r2.strings and list(r1.gaptofill) == list(r2.gaptofill)
    called = wrong = 0
    for s, t in zip(r1.strings, truth):
        t = t.tobytes().decode()
        if len(s) == len(t):
            called += sum(1 for a in s if a != "N")
            wrong += sum(1 for a, b in zip(s, t) if a != "N" and a != b)
    assert called > 0 and wrong <= 0.002 * called, (called, wrong)
    assert all(set(s) <= set("ACGTN") for s in r1.strings)
    assert all(0 <= n for n in r1.filled_len)


def test_bench_shape_matches_oracle_on_a_sample(tmp_path):
    """At bench shape the full batch is too costly for the CPU, so a sample of its gaps is written back to
    files and checked against the oracle byte for byte."""
    spec = synth.BenchSpec(mode="unmapped", reads_per_gap_mean=30.0)
    eng, mc = _bench_engine(spec)
    # index 60: a 170-bp gap (candidates up to 510 columns: the 512-thread class, two waves per read)
    batch, _ = synth.make_bench_batch(321, 64, spec, gap_lengths=np.array(([12, 40, 90, 500] * 15 + [170, 40, 90, 500])))
    res = eng.fill(batch)
    eng.close()
    sample = [0, 1, 2, 3, 9, 18, 60]
    paths = synth.write_batch_subset(batch, sample, mc, str(tmp_path / "cpu"), spec)
    case_args = [paths["scf"], str(spec.max_distance), str(spec.read_len), "1", "0", "1", "1", paths["myout"], paths["tmp"], paths["gaps"],
                 "30", str(mc.partial_len), "10", "0", str(int(spec.insert_mean))]
    r = util.run([util.ORACLE, "fillgaps"] + case_args, str(tmp_path), timeout=900)
    assert r.returncode == 0, r.stderr
    lines = util.read(paths["tmp"] + "gapout.txt").splitlines()
    order = paths["gap_order"]
    for k, g in enumerate(order):
        if g not in sample:
            continue
        f = lines[k].split("\t")
        assert int(f[4]) == int(res.filled_len[g]), f"gap {g}"
        assert (f[5] if len(f) > 5 else "") == res.strings[g], f"gap {g}"


def test_partial_mode_bench_shape_properties():
    spec = synth.BenchSpec(mode="partial", read_len=101, insert_mean=180, insert_sd=10, partial_cov=40)
    eng, _ = _bench_engine(spec)
    batch, truth = synth.make_bench_batch(77, 512, spec)
    r1 = eng.fill(batch)
    r2 = eng.fill(batch)
    eng.close()
    assert r1.strings == r2.strings
    called = wrong = 0
    for s, t in zip(r1.strings, truth):
        t = t.tobytes().decode()
        if len(s) == len(t):
            called += sum(1 for a in s if a != "N")
            wrong += sum(1 for a, b in zip(s, t) if a != "N" and a != b)
    assert called > 0 and wrong <= 0.01 * called, (called, wrong)


def test_scheduler_modes_agree(monkeypatch):
    """The candidate-parallel scheduler (speculative candidates + ordered replay, the default) and the plain
    one-workgroup-per-gap kernel (FIG_SCHED=seq) are the same device arithmetic in a different order of
    launches: strings, gaptofill, candidate records and likelihoods must be bit-identical."""
    spec = synth.BenchSpec(mode="unmapped", reads_per_gap_mean=25.0)
    eng, _ = _bench_engine(spec)
    batch, _ = synth.make_bench_batch(77, 96, spec, gap_lengths=np.array(([3, 12, 31, 40, 90, 160, 420, 1300] * 12)))
    eng.upload(batch)
    monkeypatch.delenv("FIG_SCHED", raising=False)
    par = eng.fill_resident(debug_cand=64)
    monkeypatch.setenv("FIG_SCHED", "seq")
    seq = eng.fill_resident(debug_cand=64)
    eng.free_batch(); eng.close()
    assert par.strings == seq.strings
    assert list(par.filled_len) == list(seq.filled_len) and list(par.gaptofill) == list(seq.gaptofill)
    assert list(par.n_place) == list(seq.n_place)
    for g in range(batch.n_gaps):
        a, b = par.cand[g], seq.cand[g]
        assert len(a) == len(b), f"gap {g}"
        for x, y in zip(a, b):
            assert tuple(x[:3]) == tuple(y[:3]), f"gap {g}"
            assert x[3] == y[3] or (np.isnan(x[3]) and np.isnan(y[3])), f"gap {g}: {x} vs {y}"


@pytest.mark.parametrize("seed", [531, 532])
def test_gpu_work_counters_match_oracle(seed, tmp_path):
    """The device's useful-work counters (placeReads calls, algorithmic flops; discarded speculation excluded) equal the
    oracle's: they are what bench.py's roofline figure is computed from."""
    from tools.fuzz_ref import mk
    from tools.compare_emu import run_one
    assert run_one(mk(seed), str(tmp_path), exe=util.FIGFILL, verbose=False, trace=True)


def _oracle_compare(batch, res, sample, mc, spec, tmp_path, timeout=1500):
    paths = synth.write_batch_subset(batch, sample, mc, str(tmp_path / "cpu"), spec)
    case_args = [paths["scf"], str(spec.max_distance), str(spec.read_len), "1", "0", "1", "1", paths["myout"], paths["tmp"], paths["gaps"],
                 "30", str(mc.partial_len), "10", "0", str(int(spec.insert_mean))]
    r = util.run([util.ORACLE, "fillgaps"] + case_args, str(tmp_path), timeout=timeout)
    assert r.returncode == 0, r.stderr
    lines = util.read(paths["tmp"] + "gapout.txt").splitlines()
    n = 0
    for k, g in enumerate(paths["gap_order"]):
        if g not in sample:
            continue
        f = lines[k].split("\t")
        assert int(f[4]) == int(res.filled_len[g]), f"gap {g} (G0={int(batch.gap_len[g])})"
        assert (f[5] if len(f) > 5 else "") == res.strings[g], f"gap {g} (G0={int(batch.gap_len[g])})"
        n += 1
    assert n == len(sample)


def test_near_cap_read_counts_match_oracle(tmp_path):
    """Gaps at the reference's 3000-reads-per-gap cap (Figbird.cpp:5763), one in the 512-thread four-row LDS class
    (<= 1216 columns) and one in the single-row LDS class (1217-1600 columns): bytes equal the oracle's."""
    spec = synth.BenchSpec(mode="unmapped", reads_per_gap_mean=1.0e6)
    eng, mc = _bench_engine(spec)
    batch, _ = synth.make_bench_batch(99, 2, spec, gap_lengths=np.array([450, 1500]))
    assert int(np.diff(batch.u_read_off).max()) > 2500
    res = eng.fill(batch)
    eng.close()
    _oracle_compare(batch, res, [0, 1], mc, spec, tmp_path)


def test_longest_gap_class_matches_oracle(tmp_path):
    """Gaps whose candidates exceed 1600 columns (fig_pack.h class table: the last class; BASELINE config 5's top bracket,
    gaps of 1601-2000 bp): 1700 and 2000 bp, plus a 1599/1601 pair either side of the class boundary."""
    spec = synth.BenchSpec(mode="unmapped", reads_per_gap_mean=500.0)
    eng, mc = _bench_engine(spec)
    batch, _ = synth.make_bench_batch(2027, 4, spec, gap_lengths=np.array([1700, 2000, 1599, 1601]))
    res = eng.fill(batch)
    eng.close()
    _oracle_compare(batch, res, [0, 1, 2, 3], mc, spec, tmp_path)


def test_gaps_wider_than_the_bench_mix_match_oracle(tmp_path):
    """Beyond BASELINE's 2000-bp top bracket the LDS-tiled class keeps 24 / 32 register accumulators per lane in its column
    pass (2049-3072 / 3073-4096 bp) and, past that, the generic column pass (accumulators in the scratch slab): one gap of
    each, bytes equal the oracle's."""
    spec = synth.BenchSpec(mode="unmapped", reads_per_gap_mean=160.0, gap_spacing=10000, scaffold_len=40000)
    eng, mc = _bench_engine(spec)
    batch, _ = synth.make_bench_batch(2031, 3, spec, gap_lengths=np.array([2600, 3500, 4500]))
    res = eng.fill(batch)
    eng.close()
    _oracle_compare(batch, res, [0, 1, 2], mc, spec, tmp_path)


def test_loguniform_mix_sample_matches_oracle(tmp_path):
    """BASELINE config 5's gap mix (length log-uniform in [50, 2000]): a seeded batch, a stratified sample of which
    (one gap per length octave, the cheapest of each for the CPU's sake) is checked against the oracle byte for byte."""
    spec = synth.BenchSpec(mode="unmapped", reads_per_gap_mean=24.0, gap_mix="loguniform")
    eng, mc = _bench_engine(spec)
    batch, _ = synth.make_bench_batch(555, 96, spec)
    res = eng.fill(batch)
    eng.close()
    G = np.asarray(batch.gap_len); nr = np.diff(batch.u_read_off)
    sample = []
    for lo, hi in [(50, 100), (100, 134), (134, 250), (250, 401), (401, 800), (800, 1217), (1217, 1601), (1601, 2001)]:
        ids = [g for g in range(batch.n_gaps) if lo <= G[g] < hi]
        if ids:
            sample.append(min(ids, key=lambda g: (nr[g] * (G[g] if G[g] <= 400 else 1), g)))
    assert len(sample) >= 7
    _oracle_compare(batch, res, sample, mc, spec, tmp_path, timeout=2400)


def test_config5_mix_at_its_own_depth(tmp_path):
    """BASELINE config 5 at its own depth: gap lengths log-uniform in [50, 2000], 256 gaps, ~1000 reads per gap (10^9 reads /
    10^6 gaps).  The whole batch: two fills are identical (determinism) and the called bases equal the synthetic truth at the
    substitution-error rate; one gap per length octave above 400 bp (one candidate length: seconds for the CPU) is compared
    with the oracle byte for byte.  The <= 400-bp regime at this depth costs the oracle 10^2-10^3 s per gap and is pinned by
    the committed bench_* goldens (same read length, insert size and depth)."""
    spec = synth.BenchSpec(mode="unmapped", reads_per_gap_mean=1000.0, gap_mix="loguniform")
    eng, mc = _bench_engine(spec)
    batch, truth = synth.make_bench_batch(5005, 256, spec)
    nr = np.diff(batch.u_read_off); G = np.asarray(batch.gap_len)
    assert 800 < float(nr.mean()) < 1200 and int(G.min()) >= 50 and int(G.max()) <= 2000
    eng.upload(batch)
    r1 = eng.fill_resident()
    r2 = eng.fill_resident()
    eng.free_batch(); eng.close()
    assert _loaded_native()
    assert r1.strings == r2.strings and list(r1.filled_len) == list(r2.filled_len) and list(r1.gaptofill) == list(r2.gaptofill)
    called = wrong = 0
    for s, t in zip(r1.strings, truth):
        t = t.tobytes().decode()
        if len(s) == len(t):
            called += sum(1 for a in s if a != "N")
            wrong += sum(1 for a, b in zip(s, t) if a != "N" and a != b)
    assert called > 20000 and wrong <= 0.002 * called, (called, wrong)
    sample = []
    for lo, hi in [(401, 800), (800, 1217), (1217, 1601), (1601, 2001)]:
        ids = [g for g in range(batch.n_gaps) if lo <= G[g] < hi and nr[g] <= 3000]
        if ids:
            sample.append(min(ids, key=lambda g: (nr[g], g)))
    assert len(sample) == 4
    _oracle_compare(batch, r1, sample, mc, spec, tmp_path, timeout=2400)
