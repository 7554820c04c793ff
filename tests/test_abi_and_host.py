"""The C-ABI shared library and the host-side mirror (no GPU compute)."""
import ctypes
import os
import re

import numpy as np
import pytest

import util
from figbird_amd import api, build as fbuild, synth


def _declared_functions():
    hdr = util.read(os.path.join(util.ROOT, "include", "figbird_hip.h"))
    names = re.findall(r"^\s*(?:int|void|int64_t|const char \*)\s*\*?\s*(fig_[a-z_]+)\s*\(", hdr, flags=re.M)
    return sorted(set(names))


def test_header_declares_the_documented_surface():
    assert _declared_functions() == sorted(api.EXPORTS)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(fbuild.LIB), "libfighip.so must be built in-tree (python -m figbird_amd.build)"
    lib = ctypes.CDLL(fbuild.LIB)
    for fn in _declared_functions():
        assert hasattr(lib, fn), fn
    assert lib.fig_version() == 1


def test_no_gpu_means_a_loud_failure_not_a_fallback():
    """Without a HIP device fig_ctx_create must fail (FIG_ENODEV); with one it must succeed.  Never silent CPU."""
    lib = api.load_library()
    ctx = ctypes.c_void_p()
    rc = lib.fig_ctx_create(0, ctypes.byref(ctx))
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if have:
        assert rc == 0
        lib.fig_ctx_destroy(ctx)
    else:
        assert rc == -2 and not ctx.value
        with pytest.raises(RuntimeError):
            api.Engine(0)


def test_struct_layouts_match_the_header():
    # sizes as the C compiler lays them out (x86-64 SysV): checked against a tiny C program
    import subprocess, tempfile
    src = '#include <stdio.h>\n#include "figbird_hip.h"\nint main(){printf("%zu %zu %zu %zu\\n",sizeof(fig_model),sizeof(fig_gap_batch),sizeof(fig_gap_results),sizeof(fig_stats));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.check_call(["gcc", "-I", os.path.join(util.ROOT, "include"), "-o", os.path.join(d, "t"), os.path.join(d, "t.c")])
        out = subprocess.check_output([os.path.join(d, "t")]).decode().split()
    assert [int(x) for x in out] == [ctypes.sizeof(api.FigModel), ctypes.sizeof(api.FigGapBatch), ctypes.sizeof(api.FigGapResults), ctypes.sizeof(api.FigStats)]


def test_host_model_equals_oracle_model(tmp_path):
    """A0 built by the shipped host code (libfighost.so) vs the oracle's MODEL trace line."""
    root = util.extract_golden("unmapped_mid_err", str(tmp_path))
    tr = str(tmp_path / "t.trace")
    assert util.run_oracle_fillgaps(root, trace=tr).returncode == 0
    _, om = util.parse_trace(tr)
    m = api.model_from_files(os.path.join(root, "scf.fa"), os.path.join(root, "tmp") + "/", os.path.join(root, "tmp", "myout.sam"),
                             partial_flag=0, unmapped_flag=1, script_itr=1, max_distance=690, read_length=50, neg_overlap=30, partial_len=50)
    assert (m.cutoff, m.Tmin, m.Tmax) == om[:3]
    assert m.stats == om[3:]
    assert len(m.e) == 50 and np.all(m.e > 0) and np.all(m.e < 1)
    assert abs(m.T.reshape(5, 5).sum(axis=1) - 1).max() < 1e-12


def test_set_inputmean_changes_the_model_as_in_the_reference(tmp_path):
    """set_inputmean = 1 (Figbird.cpp:6971-6973, used at :913): pairs on a contig no longer than the insert size stay out of the
    insert-size histogram.  The `inputmean` fixture (reference outputs made with the flag on) has such a contig: the host
    model equals the oracle's with the flag on, and differs from the model built with the flag off."""
    root = util.extract_golden("inputmean", str(tmp_path))
    a = util.meta(root)["fillgaps_argv"]
    assert a[13] == "1"
    tr = str(tmp_path / "t.trace")
    assert util.run_oracle_fillgaps(root, trace=tr).returncode == 0
    _, om = util.parse_trace(tr)
    kw = dict(partial_flag=0, unmapped_flag=1, script_itr=1, max_distance=int(a[1]), read_length=int(a[2]), neg_overlap=int(a[10]), partial_len=int(a[11]))
    on = api.model_from_files(os.path.join(root, "scf.fa"), os.path.join(root, "tmp") + "/", os.path.join(root, "tmp", "myout.sam"), setinputmean=1, isz=int(a[14]), **kw)
    off = api.model_from_files(os.path.join(root, "scf.fa"), os.path.join(root, "tmp") + "/", os.path.join(root, "tmp", "myout.sam"), setinputmean=0, isz=int(a[14]), **kw)
    assert (on.cutoff, on.Tmin, on.Tmax) == om[:3] and on.stats == om[3:]
    assert on.stats != off.stats and (on.Tmin, on.Tmax) != (off.Tmin, off.Tmax)


def test_case_to_batch_applies_parse_unmapped_orientation():
    c = synth.make_case("t", 5, "unmapped", [(3000, 30)], coverage=6, n_model_pairs=50)
    b = synth.case_to_batch(c)
    g = c.gaps[0]
    assert b.n_gaps == 1 and int(b.u_read_off[1]) == len(g.unmapped)
    for k, r in enumerate(g.unmapped):
        s = b.u_seq[b.u_seq_off[k]:b.u_seq_off[k + 1]].tobytes().decode()
        if r.anchor_reverse:
            assert s == r.mate_seq_fastq and b.u_is_reverse[k] == 0
        else:
            assert s == synth.revcomp(r.mate_seq_fastq) and b.u_is_reverse[k] == 1


def test_emulation_library_through_the_python_api(tmp_path):
    """api.Engine drives the same ctypes structs whichever library sits behind the ABI."""
    name = "partial_small"
    root = util.extract_golden(name, str(tmp_path))
    m = util.meta(root)
    a = m["fillgaps_argv"]
    model = api.model_from_files(os.path.join(root, "scf.fa"), os.path.join(root, "tmp") + "/", os.path.join(root, "tmp", "myout.sam"),
                                 partial_flag=int(a[4]), unmapped_flag=int(a[5]), script_itr=int(a[3]), max_distance=int(a[1]),
                                 read_length=int(a[2]), neg_overlap=int(a[10]), partial_len=int(a[11]))
    case = synth.make_case(name, 2, "partial", [(3000, 30), (6000, 120), (9000, 10)], insert_mean=180, insert_sd=10, coverage=30, n_model_pairs=800)
    eng = api.Engine(0, lib_path=util.EMULIB)
    eng.set_model(model)
    res = eng.fill(synth.case_to_batch(case))
    eng.close()
    exp = [ln.split("\t") for ln in util.read(os.path.join(root, "ref", "gapout.txt")).splitlines()]
    assert [int(e[4]) for e in exp] == list(res.filled_len)
    assert [e[5] if len(e) > 5 else "" for e in exp] == res.strings


def test_model_build_threads_agree(tmp_path, monkeypatch):
    """N2: the two passes over myout.sam split over host threads (mmap, ranges cut on pair / qname-group boundaries) give
    exactly the single-thread model: every accumulated quantity is an integer histogram."""
    import time
    c = synth.make_case("mt", 77, "unmapped", [(3000, 30)], coverage=2, n_model_pairs=9000, err=0.01, model_indel_rate=0.08, contig_len=14000)
    # multi-alignment groups: a few pairs repeated back to back under the same qnames (what `bowtie2 -k` writes)
    out = []
    for k in range(0, len(c.myout), 2):
        out += c.myout[k:k + 2]
        if (k // 2) % 97 == 0:
            out += c.myout[k:k + 2] * (1 + (k // 2) % 3)
    c.myout = out
    p = synth.write_case(c, str(tmp_path))
    models, times = [], []
    for th in ("1", "2", "5", "16"):
        monkeypatch.setenv("FIGFILL_THREADS", th)
        t0 = time.time()
        m = api.model_from_files(p["scf"], p["tmp"], p["myout"], partial_flag=0, unmapped_flag=1, script_itr=1, max_distance=c.max_distance,
                                 read_length=c.read_len, neg_overlap=30, partial_len=c.partial_len)
        times.append(time.time() - t0)
        models.append(m)
    a = models[0]
    for b in models[1:]:
        assert (a.Tmin, a.Tmax, a.cutoff, a.stats) == (b.Tmin, b.Tmax, b.cutoff, b.stats)
        for x, y in ((a.e, b.e), (a.ins, b.ins), (a.dele, b.dele), (a.T, b.T), (a.insd, b.insd)):
            assert np.array_equal(x, y)
    assert a.ins.max() > a.ins.min()            # the indel reads really fed inPosDist


@pytest.mark.skipif(not os.path.exists(os.path.join(os.path.dirname(util.REF_FIGBIRD), "FillGaps.out")) or not os.path.exists("/root/reference/Figbird.cpp"),
                    reason="needs oracle/_ref (the reference compiled in place) and the reference checkout")
@pytest.mark.parametrize("lens,threads", [([20, 500, 25, 600, 18, 33, 700], 5), ([500, 20, 600, 700, 30, 800, 900, 450, 25, 1000, 40, 1100], 3)])
def test_gap_deal_matches_live_reference(lens, threads, tmp_path):
    """FillGaps.cpp:456-649 deals the gaps to $num_threads worker processes (short gaps in turn, long ones by remaining
    capacity); `figfill` reproduces it (fig_host.cpp: thread_allocation) for gaploads.txt, the order of draw.txt and the
    process-level overlap_threshold, and so does the oracle.  Read-less gaps keep this quick: the reference compiles
    Figbird.cpp once per process.  (The committed `threads3` fixture pins the short-gap branch with reads.)"""
    from figbird_amd import synth
    specs, pos = [], 300
    for n in lens:
        specs.append((pos, n)); pos += n + 400
    case = synth.make_case("deal", 5, "unmapped", specs, contig_len=pos + 400, coverage=0.0, n_model_pairs=300)
    outs = {}
    for tag, exe in (("ref", [os.path.join(os.path.dirname(util.REF_FIGBIRD), "FillGaps.out")]), ("host", [util.EMU]), ("oracle", [util.ORACLE, "fillgaps"])):
        p = synth.write_case(case, str(tmp_path / tag))
        cwd = tmp_path / ("cwd_" + tag); cwd.mkdir()
        os.symlink("/root/reference/Figbird.cpp", str(cwd / "Figbird.cpp"))         # FillGaps.out shells out to g++ Figbird.cpp
        r = util.run(exe + synth.fillgaps_argv(case, p, n_threads=threads), str(cwd), timeout=600)
        assert r.returncode == 0, r.stderr
        outs[tag] = {fn: util.read(p["tmp"] + fn) for fn in ("gaploads.txt", "draw.txt", "gapout.txt", "filledContigs.fa")}
    assert outs["host"] == outs["ref"]
    assert outs["oracle"] == outs["ref"]


def _run_handle(root):
    import ctypes as C
    from figbird_amd import api
    h = api.load_host_library()
    argv = util.meta(root)["fillgaps_argv"]
    hh = h.fighost_run_open((C.c_char_p * 15)(*[a.encode() for a in argv]), C.create_string_buffer(256), 256)
    assert hh
    return h, hh


def _reach_and_presets(root, lib_path):
    """(reach bits measured by the library, presets carried along the reference's worker processes) of a golden's run."""
    import ctypes as C
    from figbird_amd import api
    cwd = os.getcwd(); os.chdir(root)
    try:
        h, hh = _run_handle(root)
        n = int(h.fighost_run_ngaps(hh))
        ids = np.arange(n, dtype=np.int64)
        cm = api.FigModel(); h.fighost_run_model(hh, C.byref(cm))
        cb = api.FigGapBatch(); su = C.c_int64(); sp = C.c_int64()
        assert h.fighost_run_shard(hh, api._p(ids, api.c_i64_p), n, C.byref(cb), C.byref(su), C.byref(sp)) == 0
        eng = api.Engine(0, lib_path=lib_path)
        eng.set_model_struct(cm)
        eng.upload_struct(cb)
        reach = eng.probe_reach().copy()
        eng.free_batch(); eng.close()
        preset = np.zeros(n, dtype=np.uint8)
        h.fighost_run_ot_presets(hh, api._p(reach, api.c_u8_p), api._p(preset, api.c_u8_p))
        h.fighost_run_close(hh)
    finally:
        os.chdir(cwd)
    return reach.tolist(), preset.tolist()


OT_EXPECT = {   # golden -> (reach, presets): tools/make_golden.py explains the two fixtures
    # gap 0 closes by a negative overlap at its first candidate, gap 1 sits 8 bp from the contig end: neither gets to :6317,
    # so gap 1 still sees overlap_threshold = 0 (a carry predicted from contig geometry alone handed it 5)
    "ot_carry": ([0, 0], [0, 0]),
    # $num_threads = 3: processes {0, 3}, {1}, {2}: gap 3 follows gap 0 in ITS process, although gaps 1 and 2 got there
    "ot_carry_t3": ([0, 1, 1, 0], [0, 0, 0, 0]),
    "threads3": ([1] * 8, [0, 0, 0, 1, 1, 1, 1, 1]),
    "neg_overlap": ([1, 0], [0, 1]),
}


@pytest.mark.parametrize("name", sorted(OT_EXPECT))
def test_overlap_threshold_carry_is_measured(name, tmp_path):
    """Figbird.cpp:103, :6298-6317: the library measures which gaps get to `overlap_threshold=5` (fig_batch_probe_reach; here the
    one-lane emulation of the same device code), the host carries the bits along the reference's worker processes."""
    root = util.extract_golden(name, str(tmp_path))
    assert _reach_and_presets(root, util.EMULIB) == OT_EXPECT[name]


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(OT_EXPECT))
def test_overlap_threshold_carry_is_measured_on_the_device(name, tmp_path):
    """The same through libfighip.so: fig_probe_kernel on the MI355X."""
    root = util.extract_golden(name, str(tmp_path))
    assert _reach_and_presets(root, None) == OT_EXPECT[name]
    assert "libfighip.so" in open("/proc/self/maps").read()


def test_library_measures_the_carry_itself_without_a_preset(tmp_path):
    """gap_ot_preset = NULL: one worker process in batch order, carry measured by the library at upload -- same strings as with
    the explicit carry, and (on `ot_carry`) the preset differs from what contig geometry alone would have predicted."""
    from tools import make_golden
    case = make_golden.make("ot_carry")
    root = util.extract_golden("ot_carry", str(tmp_path))
    cwd = os.getcwd(); os.chdir(root)
    try:
        model = api.model_from_files("scf.fa", "tmp/", "tmp/myout.sam", partial_flag=1, unmapped_flag=0, script_itr=1, max_distance=case.max_distance,
                                     read_length=case.read_len, neg_overlap=case.neg_overlap, partial_len=case.partial_len)
    finally:
        os.chdir(cwd)
    exp = [ln.split("\t") for ln in util.read(os.path.join(root, "ref", "gapout.txt")).splitlines()]
    outs = []
    for preset in (None, np.zeros(2, dtype=np.uint8)):
        b = synth.case_to_batch(case); b.gap_ot_preset = preset
        eng = api.Engine(0, lib_path=util.EMULIB); eng.set_model(model)
        res = eng.fill(b); eng.close()
        outs.append((res.filled_len.tolist(), res.gaptofill.tolist(), res.strings))
    assert outs[0] == outs[1]
    assert outs[0][0] == [int(e[4]) for e in exp] and outs[0][2] == [e[5] if len(e) > 5 else "" for e in exp]
