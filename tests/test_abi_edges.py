"""C-ABI edge behaviour: empty batches, invalid descriptors, inputs outside the supported envelope.
Runs against the emulation library on CPU and against libfighip.so on the GPU (same ctypes structs)."""
import copy
import os

import numpy as np
import pytest

from figbird_amd import api, synth
from tests import util


def _engine_and_batch(lib_path, tmp_path):
    spec = synth.BenchSpec(mode="unmapped", reads_per_gap_mean=8.0)
    mc = synth.bench_model_case(7, spec)
    p = synth.write_case(mc, str(tmp_path / "model"))
    model = api.model_from_files(p["scf"], p["tmp"], p["myout"], partial_flag=0, unmapped_flag=1, script_itr=1, max_distance=spec.max_distance,
                                 read_length=spec.read_len, neg_overlap=30, partial_len=mc.partial_len)
    batch, _ = synth.make_bench_batch(5, 4, spec, gap_lengths=np.array([8, 20, 8, 500]))
    eng = api.Engine(0, lib_path=lib_path)
    eng.set_model(model)
    return eng, batch


def _edge_checks(lib_path, tmp_path):
    eng, batch = _engine_and_batch(lib_path, tmp_path)
    # empty batch: nothing to do, no error
    empty = copy.copy(batch)
    for f in ("gap_contig", "gap_start", "gap_len", "gap_fillflag"):
        setattr(empty, f, getattr(batch, f)[:0].copy())
    empty.gap_stat2 = batch.gap_stat2[:0].copy()
    empty.u_read_off = np.zeros(1, dtype=np.int64); empty.p_read_off = np.zeros(1, dtype=np.int64)
    res = eng.fill(empty)
    assert len(res.strings) == 0 and len(res.filled_len) == 0
    # a ragged batch: gaps without any read next to gaps with reads
    few = synth.subset_batch(batch, [0, 2])
    few.u_read_off = np.array([0, 0, int(few.u_read_off[-1])], dtype=np.int64)       # gap 0 loses its reads to gap 1
    res = eng.fill(few)
    assert res.strings[0] in ("", "N" * int(few.gap_len[0])) or set(res.strings[0]) <= set("ACGTN")
    # gap outside its contig -> FIG_EINVAL, reported, not crashed
    bad = copy.copy(batch)
    bad.gap_start = batch.gap_start.copy(); bad.gap_start[0] = 10 ** 9
    with pytest.raises(RuntimeError, match="invalid argument"):
        eng.fill(bad)
    # contig index out of range -> FIG_EINVAL
    bad = copy.copy(batch)
    bad.gap_contig = batch.gap_contig.copy(); bad.gap_contig[1] = 10 ** 6
    with pytest.raises(RuntimeError, match="invalid argument"):
        eng.fill(bad)
    # a read longer than MAX_READLENGTH (Figbird.cpp:18) -> outside the envelope, FIG_EUNSUP
    bad = copy.copy(batch)
    bad.u_seq_off = batch.u_seq_off.copy(); bad.u_seq_off[1:] += 400
    bad.u_seq = np.concatenate([np.full(400, ord("A"), dtype=np.uint8), batch.u_seq])
    with pytest.raises(RuntimeError, match="supported envelope"):
        eng.fill(bad)
    # a partial (soft-clipped) read longer than the model's max_read_length would index the {1-e,e} pair tables past
    # their end -> FIG_EUNSUP (the frag-library reads of this batch are 101 bp; the model's L is 150)
    bad = copy.copy(batch)
    npr = len(batch.p_seq_off) - 1
    if npr > 0:
        bad.p_seq_off = batch.p_seq_off.copy(); bad.p_seq_off[1:] += 60        # first partial read: 161 bp > L = 150
        bad.p_seq = np.concatenate([np.full(60, ord("A"), dtype=np.uint8), batch.p_seq])
        bad.p_qual = np.concatenate([np.full(60, ord("I"), dtype=np.uint8), batch.p_qual])
        with pytest.raises(RuntimeError, match="supported envelope"):
            eng.fill(bad)
    # CSR offsets that go backwards -> FIG_EINVAL (would give a negative read count in the gap descriptor)
    bad = copy.copy(batch)
    bad.u_read_off = batch.u_read_off.copy(); bad.u_read_off[2] = bad.u_read_off[1] - 1 if bad.u_read_off[1] > 0 else -1
    with pytest.raises(RuntimeError, match="invalid argument"):
        eng.fill(bad)
    bad = copy.copy(batch)
    bad.p_read_off = batch.p_read_off.copy(); bad.p_read_off[0] = -3
    with pytest.raises(RuntimeError, match="invalid argument"):
        eng.fill(bad)
    # a new model drops the resident batch (it was packed under the old one): fill_resident must refuse, not run stale
    eng.upload(batch)
    eng.set_model(eng._model)
    with pytest.raises(RuntimeError, match="invalid argument"):
        eng.fill_resident()
    # the engine is still usable after the errors and gives the same answer as before them
    r1 = eng.fill(batch)
    r2 = eng.fill(batch)
    assert r1.strings == r2.strings and list(r1.filled_len) == list(r2.filled_len)
    eng.close()


def test_abi_edges_on_the_emulation_library(tmp_path):
    _edge_checks(util.EMULIB, tmp_path)


@pytest.mark.gpu
def test_abi_edges_on_the_device(tmp_path):
    _edge_checks(None, tmp_path)
