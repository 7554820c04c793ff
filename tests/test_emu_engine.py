"""CPU checks of the PRODUCT's host code and engine control logic without a GPU: figfill's host side
(file ingest, model, packer, writers) is the shipped code; the per-gap engine (figbird_amd/csrc/fig_engine*.h)
is compiled for the host as a one-lane emulation (tests/emu, test infrastructure only).  Parity proper --
the real HIP kernels through the C ABI -- is in test_gpu_parity.py."""
import os

import pytest

import util


@pytest.mark.parametrize("name", util.GOLDEN_CASES)
def test_emulated_engine_matches_reference_outputs(name, tmp_path):
    root = util.extract_golden(name, str(tmp_path))
    r = util.run_figfill(root, util.EMU)
    assert r.returncode == 0, r.stderr
    for fn in util.ref_files(root):
        assert util.read(os.path.join(root, "tmp", fn)) == util.read(os.path.join(root, "ref", fn)), fn


@pytest.mark.parametrize("name", ["unmapped_small", "partial_brackets", "neg_overlap"])
def test_candidate_planes_match_oracle(name, tmp_path):
    """likelihood_arr (Figbird.cpp:6390-6391): per candidate gap length the EM iteration count, valid_count and
    likelihood; same libm on both sides here, so they must agree exactly."""
    a = util.extract_golden(name, str(tmp_path / "a"))
    b = util.extract_golden(name, str(tmp_path / "b"))
    ta, tb = str(tmp_path / "ora.trace"), str(tmp_path / "emu.trace")
    assert util.run_oracle_fillgaps(a, trace=ta).returncode == 0
    assert util.run_figfill(b, util.EMU, trace=tb).returncode == 0
    ca, ma = util.parse_trace(ta)
    cb, mb = util.parse_trace(tb)
    assert ma == mb                      # host model (figfill) == oracle model: cutoff, thresholds, mean, SDs
    assert ca.keys() == cb.keys()
    for g in ca:
        assert ca[g] == cb[g], f"gap {g}"


@pytest.mark.parametrize("seed", [401, 402, 403, 404, 405, 406, 407, 408])
def test_emulated_engine_fuzz(seed, tmp_path):
    from tools.fuzz_ref import mk
    from tools.compare_emu import run_one
    assert run_one(mk(seed), str(tmp_path), exe=util.EMU, verbose=False)


@pytest.mark.parametrize("chunk", ["1", "3", "64"])
def test_emulated_scheduler_chunking_is_invariant(chunk, tmp_path, monkeypatch):
    """The candidate-parallel orchestration (speculate CHUNK candidates, replay the bookkeeping in order) must give
    the reference's bytes whatever the chunk size; FIG_SCHED=seq (one pass per gap) is covered by the other tests'
    default path only on the GPU, so it is pinned here too."""
    from tools.fuzz_ref import mk
    from tools.compare_emu import run_one
    monkeypatch.setenv("FIG_EMU_CHUNK", chunk)
    for seed in (411, 412):
        assert run_one(mk(seed), str(tmp_path / f"c{seed}"), exe=util.EMU, verbose=False)
    monkeypatch.setenv("FIG_SCHED", "seq")
    assert run_one(mk(413), str(tmp_path / "seq"), exe=util.EMU, verbose=False)


@pytest.mark.parametrize("seed", [421, 422, 423])
def test_work_counters_match_oracle(seed, tmp_path):
    """placeReads calls and algorithmic flops (4 per E-step base, 1 per MLE base, 1 per countsGap add: SURVEY §8d) are
    counted by the oracle and by the engine; bench.py's roofline uses the engine's, so they must be the same number."""
    from tools.fuzz_ref import mk
    from tools.compare_emu import run_one
    assert run_one(mk(seed), str(tmp_path), exe=util.EMU, verbose=False, trace=True)


@pytest.mark.parametrize("tiles", ["2", "3", "5"])
def test_emulated_lds_tiled_estep(tiles, tmp_path, monkeypatch):
    """The >1600-column class streams the {P,Q} table through an LDS image one column tile at a time
    (fig_hot_estep<.., TILED>).  FIG_EMU_TILES forces that form on every class, so the tile bookkeeping (placement ->
    tile assignment, sub-windows, unit dealing, per-read arg-max slots) is compared with the reference's bytes on
    small gaps too; the real tile sizes run in test_gpu_parity.py::test_longest_gap_class_matches_oracle."""
    from tools.fuzz_ref import mk
    from tools.compare_emu import run_one
    monkeypatch.setenv("FIG_EMU_TILES", tiles)
    for name in ("unmapped_small", "repeat_flanks", "model_indels"):
        root = util.extract_golden(name, str(tmp_path / name))
        r = util.run_figfill(root, util.EMU)
        assert r.returncode == 0, r.stderr
        assert f"LDS-tiled E-step forced, tiles={tiles} " in r.stderr          # the knob reached the emulation
        for fn in ("gapout.txt", "filledContigs.fa", "Ncount.txt", "draw.txt"):
            assert util.read(os.path.join(root, "tmp", fn)) == util.read(os.path.join(root, "ref", fn)), (name, fn)
    for seed in (431, 432):
        assert run_one(mk(seed), str(tmp_path / f"f{seed}"), exe=util.EMU, verbose=False, trace=True)


def test_emulated_split_class_lanes(tmp_path):
    """A class with >= FIG_SPLIT_MIN_GAPS (96) gaps is scheduled as two lanes of the same memory form (fig_pack.h): 120
    small gaps through the shipped host code + packer + the emulated engine give the oracle's bytes, and the packer did
    split (FIGFILL_TRACE lists the launch classes)."""
    import numpy as np
    from figbird_amd import synth
    spec = synth.BenchSpec(mode="unmapped", reads_per_gap_mean=3.0, read_len=60, insert_mean=900.0, insert_sd=60.0, frag_len=60)
    mc = synth.bench_model_case(7, spec)
    rng = np.random.default_rng(97)
    batch, _ = synth.make_bench_batch(97, 120, spec, gap_lengths=rng.integers(5, 16, size=120))
    outs = {}
    for who, exe in (("oracle", None), ("emu", util.EMU)):
        paths = synth.write_batch_subset(batch, list(range(120)), mc, str(tmp_path / who), spec)
        args = [paths["scf"], str(spec.max_distance), str(spec.read_len), "1", "0", "1", "1", paths["myout"], paths["tmp"], paths["gaps"],
                "30", str(mc.partial_len), "10", "0", str(int(spec.insert_mean))]
        env = dict(os.environ)
        if exe:
            env["FIGFILL_TRACE"] = str(tmp_path / "emu.trace")
        r = util.run(([util.ORACLE, "fillgaps"] if exe is None else [exe]) + args, str(tmp_path), timeout=900, env=env)
        assert r.returncode == 0, r.stderr
        outs[who] = {fn: util.read(paths["tmp"] + fn) for fn in ("gapout.txt", "filledContigs.fa", "Ncount.txt")}
    assert outs["emu"] == outs["oracle"]
    tr = util.read(str(tmp_path / "emu.trace"))
    lanes = [int(ln.split("\t")[1]) for ln in tr.splitlines() if ln.startswith("LAUNCHES\t")]
    assert lanes and lanes[0] >= 2, "expected the one class to be split into two lanes"
