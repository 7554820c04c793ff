"""Numeric planes (i) and (ii) of the parity contract (SURVEY.md §8 preamble): per candidate length, countsGap[x][0..4]
as the last E-step left it and the per-read E-step maxima (maxlikelihood_value), exported through the C ABI's debug
planes and compared with the oracle's level-3 trace.  Tolerance 1e-6 relative (device libm vs glibc differ in the last
ulp; the emulation build uses glibc and the reference's summation order, so it must be exact); 0 and +-inf exact."""
import os

import numpy as np
import pytest

import util
from figbird_amd import api, synth


def parse_planes(path):
    """-> {gap: [(gapEstimate, counts[G,5], rmax[R])]} : the last E / R records before each CAND line."""
    out, last_e, last_r = {}, {}, {}
    for ln in open(path):
        f = ln.rstrip("\n").split("\t")
        if f[0] == "E":
            last_e[int(f[1])] = (int(f[2]), np.array([float.fromhex(x) for x in f[4:]]).reshape(-1, 5))
        elif f[0] == "R":
            last_r[int(f[1])] = (int(f[2]), np.array([float.fromhex(x) for x in f[4:]]))
        elif f[0] == "CAND":
            g, G = int(f[1]), int(f[2])
            e = last_e.get(g); r = last_r.get(g)
            out.setdefault(g, []).append((G, e[1] if e and e[0] == G else None, r[1] if r and r[0] == G else None))
            last_e.pop(g, None); last_r.pop(g, None)
    return out


def _close(a, b, tol):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    special = ~np.isfinite(a) | ~np.isfinite(b) | (a == 0) | (b == 0)
    ok = np.where(special, (a == b) | (np.isnan(a) & np.isnan(b)), np.abs(a - b) <= tol * np.maximum(np.abs(a), np.abs(b)))
    return bool(ok.all()), (int(np.argmin(ok)) if not ok.all() else -1)


def check_planes(lib_path, name, tmp_path, tol):
    from tools.make_golden import make
    root = util.extract_golden(name, str(tmp_path))
    tr = str(tmp_path / "o.trace")
    assert util.run_oracle_fillgaps(root, trace=tr, level=3).returncode == 0
    planes = parse_planes(tr)
    case = make(name)
    a = util.meta(root)["fillgaps_argv"]
    model = api.model_from_files(os.path.join(root, "scf.fa"), os.path.join(root, "tmp") + "/", os.path.join(root, "tmp", "myout.sam"),
                                 partial_flag=int(a[4]), unmapped_flag=int(a[5]), script_itr=int(a[3]), max_distance=int(a[1]),
                                 read_length=int(a[2]), neg_overlap=int(a[10]), partial_len=int(a[11]))
    batch = synth.case_to_batch(case)
    off = batch.u_read_off if case.mode == "unmapped" else batch.p_read_off
    cols = int(max(max(G for G, _, _ in v) for v in planes.values())) + 1
    reads = int(np.diff(off).max()) + 1
    eng = api.Engine(0, lib_path=lib_path)
    eng.set_model(model)
    res = eng.fill(batch, debug_cand=512, plane_cols=cols, plane_reads=reads)
    eng.close()
    n_checked = 0
    for g, recs in planes.items():
        got = res.cand[g]
        assert len(got) == len(recs), f"gap {g}: candidate count"
        for k, ((G, cnt, rmax), gc) in enumerate(zip(recs, got)):
            assert gc[0] == G, f"gap {g} candidate {k}"
            if cnt is not None:
                ok, at = _close(res.counts[g, k, :G, :].reshape(-1), cnt.reshape(-1), tol)
                assert ok, f"gap {g} G={G}: countsGap column {at // 5} base {at % 5}: {res.counts[g, k, at // 5, at % 5]!r} vs {cnt.reshape(-1)[at]!r}"
                n_checked += 1
            if rmax is not None:
                ok, at = _close(res.read_maxlv[g, k, :len(rmax)], rmax, tol)
                assert ok, f"gap {g} G={G}: read {at}: {res.read_maxlv[g, k, at]!r} vs {rmax[at]!r}"
    assert n_checked > 0
    return n_checked


@pytest.mark.parametrize("name", ["unmapped_small", "partial_small"])
def test_planes_emulation_equals_oracle_exactly(name, tmp_path):
    """Host emulation of the device engine (glibc math, same summation order): planes bit-identical to the oracle."""
    check_planes(util.EMULIB, name, tmp_path, tol=0.0)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["unmapped_small", "unmapped_mid_err", "partial_small", "partial_brackets"])
def test_planes_on_gpu_within_1e6(name, tmp_path):
    check_planes(None, name, tmp_path, tol=1e-6)
