"""ctypes binding of libfighip.so (include/figbird_hip.h) and libfighost.so.

This is the Python face of the C ABI used by the tests and bench.py.  It does no computing
itself: every gap is filled by the HIP engine behind `fig_fill_gaps`.  If the library or a
GPU is missing it raises -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from . import build as _build

c_double_p = C.POINTER(C.c_double)
c_i32_p = C.POINTER(C.c_int32)
c_i64_p = C.POINTER(C.c_int64)
c_u8_p = C.POINTER(C.c_uint8)


class FigModel(C.Structure):
    _fields_ = [
        ("max_read_length", C.c_int32), ("error_pos_dist", c_double_p), ("in_pos_dist", c_double_p),
        ("del_pos_dist", c_double_p), ("error_type_probs", C.c_double * 25),
        ("insert_len_dist_smoothed", c_double_p), ("max_insert_size", C.c_int32),
        ("insert_threshold_min", C.c_int32), ("insert_threshold_max", C.c_int32), ("gap_prob_cutoff", C.c_int32),
        ("partial_flag", C.c_int32), ("unmapped_flag", C.c_int32), ("script_itr", C.c_int32),
        ("max_distance", C.c_int32), ("read_length", C.c_int32), ("neg_overlap", C.c_int32),
        ("partial_len", C.c_int32), ("unm_limit", C.c_int32),
    ]


class FigGapBatch(C.Structure):
    _fields_ = [
        ("n_gaps", C.c_int64), ("n_contigs", C.c_int64), ("contig_off", c_i64_p), ("contig_seq", C.c_char_p),
        ("gap_contig", c_i32_p), ("gap_start", c_i64_p), ("gap_len", c_i32_p), ("gap_stat2", c_i32_p),
        ("gap_fillflag", c_i32_p),
        ("u_read_off", c_i64_p), ("u_anchor_pos", c_i32_p), ("u_is_reverse", c_u8_p), ("u_seq_off", c_i64_p),
        ("u_seq", C.c_char_p),
        ("p_read_off", c_i64_p), ("p_clipped_index", c_i32_p), ("p_match", c_i32_p), ("p_pos", c_i32_p),
        ("p_ref_pos", c_i32_p), ("p_seq_off", c_i64_p), ("p_seq", C.c_char_p), ("p_qual", C.c_char_p),
        ("gap_ot_preset", c_u8_p),
    ]


class FigGapResults(C.Structure):
    _fields_ = [
        ("filled_len", c_i32_p), ("gaptofill", c_i32_p), ("str_off", c_i64_p), ("str", C.c_char_p),
        ("str_capacity", C.c_int64),
        ("dbg_max_cand", C.c_int32), ("dbg_n_cand", c_i32_p), ("dbg_cand_i", c_i32_p), ("dbg_cand_lik", c_double_p),
        ("dbg_n_place", c_i32_p),
        ("draw_pos", c_i32_p), ("draw_isz", c_i32_p), ("draw_len", c_i32_p),
        ("dbg_plane_cols", C.c_int32), ("dbg_plane_reads", C.c_int32), ("dbg_counts", c_double_p), ("dbg_read_maxlv", c_double_p),
    ]


class FigStats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("h2d_ms", C.c_double), ("d2h_ms", C.c_double),
                ("packed_bytes", C.c_int64), ("place_calls", C.c_int64), ("alg_flops", C.c_double),
                ("n_launches", C.c_int32), ("pad0", C.c_int32), ("spec_flops", C.c_double),
                ("mle_alg_flops", C.c_double), ("mle_exec_flops", C.c_double)]


EXPORTS = ["fig_version", "fig_strerror", "fig_ctx_create", "fig_ctx_destroy", "fig_ctx_set_model",
           "fig_results_capacity", "fig_batch_upload", "fig_fill_resident", "fig_batch_free", "fig_fill_gaps",
           "fig_get_stats", "fig_batch_probe_reach", "fig_batch_set_ot_preset"]

_lib = None
_host = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    """dlopen libfighip.so and set prototypes.  Raises if it is missing (no fallback)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or _build.LIB
    if not os.path.exists(p):
        raise RuntimeError(f"libfighip.so not built at {p}: run `python -m figbird_amd.build` (there is no CPU fallback)")
    lib = C.CDLL(p)
    lib.fig_version.restype = C.c_int
    lib.fig_strerror.restype = C.c_char_p
    lib.fig_strerror.argtypes = [C.c_int]
    lib.fig_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.fig_ctx_destroy.argtypes = [C.c_void_p]
    lib.fig_ctx_destroy.restype = None
    lib.fig_ctx_set_model.argtypes = [C.c_void_p, C.POINTER(FigModel)]
    lib.fig_results_capacity.argtypes = [C.POINTER(FigModel), C.POINTER(FigGapBatch)]
    lib.fig_results_capacity.restype = C.c_int64
    lib.fig_batch_upload.argtypes = [C.c_void_p, C.POINTER(FigGapBatch)]
    lib.fig_fill_resident.argtypes = [C.c_void_p, C.POINTER(FigGapResults)]
    lib.fig_batch_free.argtypes = [C.c_void_p]
    lib.fig_batch_free.restype = None
    lib.fig_fill_gaps.argtypes = [C.c_void_p, C.POINTER(FigGapBatch), C.POINTER(FigGapResults)]
    lib.fig_get_stats.argtypes = [C.c_void_p, C.POINTER(FigStats)]
    lib.fig_batch_probe_reach.argtypes = [C.c_void_p, c_u8_p]
    lib.fig_batch_set_ot_preset.argtypes = [C.c_void_p, c_u8_p]
    if path is None:
        _lib = lib
    return lib


def load_host_library() -> C.CDLL:
    global _host
    if _host is None:
        if not os.path.exists(_build.HOSTLIB):
            raise RuntimeError("libfighost.so not built: run `python -m figbird_amd.build`")
        _host = C.CDLL(_build.HOSTLIB)
        _host.fighost_build_model.restype = C.c_int
        _host.fighost_run_open.restype = C.c_void_p
        _host.fighost_run_open.argtypes = [C.POINTER(C.c_char_p), C.c_char_p, C.c_int]
        _host.fighost_run_close.argtypes = [C.c_void_p]; _host.fighost_run_close.restype = None
        _host.fighost_run_ngaps.argtypes = [C.c_void_p]; _host.fighost_run_ngaps.restype = C.c_int64
        _host.fighost_run_sizes.argtypes = [C.c_void_p, c_i32_p, c_i64_p, c_i64_p]
        _host.fighost_run_params.argtypes = [C.c_void_p, c_i32_p]
        _host.fighost_run_message.argtypes = [C.c_void_p, C.c_int]; _host.fighost_run_message.restype = C.c_char_p
        _host.fighost_run_model.argtypes = [C.c_void_p, C.POINTER(FigModel)]
        _host.fighost_run_shard.argtypes = [C.c_void_p, c_i64_p, C.c_int64, C.POINTER(FigGapBatch), c_i64_p, c_i64_p]
        _host.fighost_run_ot_presets.argtypes = [C.c_void_p, c_u8_p, c_u8_p]
        _host.fighost_run_write.argtypes = [C.c_void_p, c_i32_p, c_i32_p, c_i64_p, C.c_char_p, c_i32_p, c_i32_p, c_i32_p, C.c_char_p, C.c_int]
    return _host


def _p(arr, typ):
    return arr.ctypes.data_as(typ)


@dataclass
class Model:
    """Run-level model tables (A0) + run parameters; mirrors `fig_model`."""
    e: np.ndarray
    ins: np.ndarray
    dele: np.ndarray
    T: np.ndarray
    insd: np.ndarray
    Tmin: int
    Tmax: int
    cutoff: int
    partial_flag: int
    unmapped_flag: int
    script_itr: int
    max_distance: int
    read_length: int
    neg_overlap: int
    partial_len: int
    unm_limit: int = 400
    stats: tuple = (0.0, 0.0, 0.0)

    def cstruct(self) -> FigModel:
        m = FigModel()
        m.max_read_length = len(self.e)
        m.error_pos_dist = _p(self.e, c_double_p)
        m.in_pos_dist = _p(self.ins, c_double_p)
        m.del_pos_dist = _p(self.dele, c_double_p)
        for i in range(25):
            m.error_type_probs[i] = float(self.T[i])
        m.insert_len_dist_smoothed = _p(self.insd, c_double_p)
        m.max_insert_size = len(self.insd)
        m.insert_threshold_min = self.Tmin
        m.insert_threshold_max = self.Tmax
        m.gap_prob_cutoff = self.cutoff
        m.partial_flag = self.partial_flag
        m.unmapped_flag = self.unmapped_flag
        m.script_itr = self.script_itr
        m.max_distance = self.max_distance
        m.read_length = self.read_length
        m.neg_overlap = self.neg_overlap
        m.partial_len = self.partial_len
        m.unm_limit = self.unm_limit
        return m


def model_from_files(scf: str, tmp_dir: str, map_file: str, *, partial_flag: int, unmapped_flag: int, script_itr: int,
                     max_distance: int, read_length: int, neg_overlap: int, partial_len: int,
                     setinputmean: int = 0, isz: int = 0) -> Model:
    """Build the model exactly as figfill does (host C++, figbird_amd/csrc/host/fig_host.cpp)."""
    h = load_host_library()
    cap_l, cap_i = 256, 1 << 20
    e = np.zeros(cap_l); ins = np.zeros(cap_l); dele = np.zeros(cap_l); T = np.zeros(25); insd = np.zeros(cap_i)
    ints = np.zeros(5, dtype=np.int32); st = np.zeros(3)
    rc = h.fighost_build_model(scf.encode(), tmp_dir.encode(), map_file.encode(), C.c_int(partial_flag), C.c_int(partial_len),
                               C.c_int(setinputmean), C.c_int(isz), C.c_int(cap_l), C.c_int(cap_i),
                               _p(e, c_double_p), _p(ins, c_double_p), _p(dele, c_double_p), _p(T, c_double_p),
                               _p(insd, c_double_p), _p(ints, c_i32_p), _p(st, c_double_p))
    if rc != 0:
        raise RuntimeError(f"fighost_build_model failed rc={rc}")
    L, mi = int(ints[0]), int(ints[1])
    return Model(e=e[:L].copy(), ins=ins[:L].copy(), dele=dele[:L].copy(), T=T, insd=insd[:mi].copy(), Tmin=int(ints[2]),
                 Tmax=int(ints[3]), cutoff=int(ints[4]), partial_flag=partial_flag, unmapped_flag=unmapped_flag,
                 script_itr=script_itr, max_distance=max_distance, read_length=read_length, neg_overlap=neg_overlap,
                 partial_len=partial_len, stats=(float(st[0]), float(st[1]), float(st[2])))


@dataclass
class GapBatch:
    """Host arrays behind `fig_gap_batch` (SoA + CSR)."""
    contig_off: np.ndarray
    contig_seq: np.ndarray          # uint8 ASCII, upper case
    gap_contig: np.ndarray
    gap_start: np.ndarray
    gap_len: np.ndarray
    gap_stat2: np.ndarray
    gap_fillflag: np.ndarray
    u_read_off: np.ndarray
    u_anchor_pos: np.ndarray
    u_is_reverse: np.ndarray
    u_seq_off: np.ndarray
    u_seq: np.ndarray
    p_read_off: np.ndarray
    p_clipped_index: np.ndarray
    p_match: np.ndarray
    p_pos: np.ndarray
    p_ref_pos: np.ndarray
    p_seq_off: np.ndarray
    p_seq: np.ndarray
    p_qual: np.ndarray
    gap_ot_preset: Optional[np.ndarray] = None      # uint8 [n_gaps]; None = one reference process in batch order (figbird_hip.h)

    @property
    def n_gaps(self) -> int:
        return len(self.gap_len)

    def cstruct(self) -> FigGapBatch:
        b = FigGapBatch()
        b.n_gaps = self.n_gaps
        b.n_contigs = len(self.contig_off) - 1
        b.contig_off = _p(self.contig_off, c_i64_p)
        b.contig_seq = C.cast(self.contig_seq.ctypes.data, C.c_char_p)
        b.gap_contig = _p(self.gap_contig, c_i32_p)
        b.gap_start = _p(self.gap_start, c_i64_p)
        b.gap_len = _p(self.gap_len, c_i32_p)
        b.gap_stat2 = _p(self.gap_stat2, c_i32_p)
        b.gap_fillflag = _p(self.gap_fillflag, c_i32_p)
        b.u_read_off = _p(self.u_read_off, c_i64_p)
        b.u_anchor_pos = _p(self.u_anchor_pos, c_i32_p)
        b.u_is_reverse = _p(self.u_is_reverse, c_u8_p)
        b.u_seq_off = _p(self.u_seq_off, c_i64_p)
        b.u_seq = C.cast(self.u_seq.ctypes.data, C.c_char_p)
        b.p_read_off = _p(self.p_read_off, c_i64_p)
        b.p_clipped_index = _p(self.p_clipped_index, c_i32_p)
        b.p_match = _p(self.p_match, c_i32_p)
        b.p_pos = _p(self.p_pos, c_i32_p)
        b.p_ref_pos = _p(self.p_ref_pos, c_i32_p)
        b.p_seq_off = _p(self.p_seq_off, c_i64_p)
        b.p_seq = C.cast(self.p_seq.ctypes.data, C.c_char_p)
        b.p_qual = C.cast(self.p_qual.ctypes.data, C.c_char_p)
        b.gap_ot_preset = _p(self.gap_ot_preset, c_u8_p) if self.gap_ot_preset is not None else None
        return b


@dataclass
class FillResult:
    filled_len: np.ndarray
    gaptofill: np.ndarray
    strings: List[str]
    cand: Optional[list] = None     # per gap: list of (gapEstimate, iterations, valid_count, likelihood)
    n_place: Optional[np.ndarray] = None
    counts: Optional[np.ndarray] = None    # plane (i): [n_gaps, debug_cand, plane_cols, 5] countsGap after the candidate's last E-step
    read_maxlv: Optional[np.ndarray] = None  # plane (ii): [n_gaps, debug_cand, plane_reads] per-read E-step maximum
    str_off: Optional[np.ndarray] = None   # int64[n+1]: gap g's string is raw[str_off[g]:str_off[g+1]]
    raw: Optional[np.ndarray] = None       # uint8: all gap strings back to back, as the C ABI wrote them
    draw: Optional[tuple] = None           # (draw_pos, draw_isz, draw_len) planes of fig_gap_results, when requested

    @property
    def filled_bases(self) -> int:
        return int(sum(len(s) - s.count("N") for s in self.strings))


class Engine:
    """One `fig_ctx` (= one GPU)."""

    def __init__(self, device: int = 0, lib_path: Optional[str] = None):
        self.lib = load_library(lib_path)
        self.ctx = C.c_void_p()
        rc = self.lib.fig_ctx_create(device, C.byref(self.ctx))
        if rc != 0:
            raise RuntimeError(f"fig_ctx_create failed: {self.lib.fig_strerror(rc).decode()} (no CPU fallback)")
        self._keep = []
        self.n_gaps = 0
        self.cap = 0

    def close(self):
        if self.ctx:
            self.lib.fig_ctx_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed: {self.lib.fig_strerror(rc).decode()}")

    def set_model(self, model: Model):
        self._model = model
        self._cm = model.cstruct()
        self._check(self.lib.fig_ctx_set_model(self.ctx, C.byref(self._cm)), "fig_ctx_set_model")

    def set_model_struct(self, cm: "FigModel"):
        """Model given as a ready `fig_model` (pointers owned by the caller, e.g. libfighost's run handle)."""
        self._model = None
        self._cm = cm
        self._check(self.lib.fig_ctx_set_model(self.ctx, C.byref(self._cm)), "fig_ctx_set_model")

    def upload(self, batch: GapBatch):
        self._batch = batch
        self._cb = batch.cstruct()
        self.n_gaps = batch.n_gaps
        self.cap = int(self.lib.fig_results_capacity(C.byref(self._cm), C.byref(self._cb)))
        self._check(self.lib.fig_batch_upload(self.ctx, C.byref(self._cb)), "fig_batch_upload")

    def fill_resident(self, debug_cand: int = 0, plane_cols: int = 0, plane_reads: int = 0) -> FillResult:
        n = self.n_gaps
        fl = np.zeros(max(n, 1), dtype=np.int32); gt = np.zeros(max(n, 1), dtype=np.int32)
        so = np.zeros(n + 1, dtype=np.int64); st = np.zeros(max(self.cap, 1), dtype=np.uint8)
        r = FigGapResults()
        r.filled_len = _p(fl, c_i32_p); r.gaptofill = _p(gt, c_i32_p); r.str_off = _p(so, c_i64_p)
        r.str = C.cast(st.ctypes.data, C.c_char_p); r.str_capacity = len(st)
        if debug_cand > 0:
            dn = np.zeros(max(n, 1), dtype=np.int32); di = np.zeros(max(n, 1) * debug_cand * 3, dtype=np.int32)
            dl = np.zeros(max(n, 1) * debug_cand)
            dp = np.zeros(max(n, 1), dtype=np.int32)
            r.dbg_max_cand = debug_cand; r.dbg_n_cand = _p(dn, c_i32_p); r.dbg_cand_i = _p(di, c_i32_p); r.dbg_cand_lik = _p(dl, c_double_p); r.dbg_n_place = _p(dp, c_i32_p)
            if plane_cols > 0:
                pc = np.zeros((max(n, 1), debug_cand, plane_cols, 5))
                r.dbg_plane_cols = plane_cols; r.dbg_counts = _p(pc, c_double_p)
            if plane_reads > 0:
                pr = np.zeros((max(n, 1), debug_cand, plane_reads))
                r.dbg_plane_reads = plane_reads; r.dbg_read_maxlv = _p(pr, c_double_p)
        self._check(self.lib.fig_fill_resident(self.ctx, C.byref(r)), "fig_fill_resident")
        raw = st.tobytes()
        strings = [raw[so[g]:so[g + 1]].decode() for g in range(n)]
        cand = None
        if debug_cand > 0:
            cand = []
            for g in range(n):
                k = min(int(dn[g]), debug_cand)
                base = g * debug_cand
                cand.append([(int(di[(base + j) * 3]), int(di[(base + j) * 3 + 1]), int(di[(base + j) * 3 + 2]), float(dl[base + j])) for j in range(k)])
        res = FillResult(fl[:n].copy(), gt[:n].copy(), strings, cand, str_off=so, raw=st)
        res.n_place = dp[:n].copy() if debug_cand > 0 else None
        if debug_cand > 0 and plane_cols > 0:
            res.counts = pc[:n]
        if debug_cand > 0 and plane_reads > 0:
            res.read_maxlv = pr[:n]
        return res

    def free_batch(self):
        self.lib.fig_batch_free(self.ctx)

    def probe_reach(self) -> np.ndarray:
        """uint8[n_gaps] of the resident batch: 1 = the gap's candidate loop gets to Figbird.cpp:6317 (fig_batch_probe_reach)."""
        r = np.zeros(max(self.n_gaps, 1), dtype=np.uint8)
        if self.n_gaps > 0:
            self._check(self.lib.fig_batch_probe_reach(self.ctx, _p(r, c_u8_p)), "fig_batch_probe_reach")
        return r[:self.n_gaps]

    def set_ot_preset(self, preset: np.ndarray):
        """Replace `gap_ot_preset` of the resident batch (fig_batch_set_ot_preset)."""
        a = np.ascontiguousarray(preset, dtype=np.uint8)
        if len(a) != self.n_gaps:
            raise ValueError("set_ot_preset: one entry per gap of the resident batch")
        if self.n_gaps > 0:
            self._check(self.lib.fig_batch_set_ot_preset(self.ctx, _p(a, c_u8_p)), "fig_batch_set_ot_preset")

    def upload_struct(self, cbatch: "FigGapBatch"):
        """fig_batch_upload of a caller-built `fig_gap_batch` (e.g. a shard view from libfighost's run handle)."""
        self._cb = cbatch
        self.n_gaps = int(cbatch.n_gaps)
        self.cap = int(self.lib.fig_results_capacity(C.byref(self._cm), C.byref(cbatch))) if self.n_gaps > 0 else 1
        if self.n_gaps > 0:
            self._check(self.lib.fig_batch_upload(self.ctx, C.byref(cbatch)), "fig_batch_upload")

    def fill_struct(self, cbatch: "FigGapBatch", n_ureads: int, n_preads: int, draw: bool = True, resident: bool = False) -> FillResult:
        """fig_fill_gaps on a caller-built `fig_gap_batch` (e.g. a shard view from libfighost's run handle), with the
        per-read draw planes; returns a FillResult whose `draw` field holds (draw_pos, draw_isz, draw_len).
        resident=True: the batch was uploaded with upload_struct (fig_fill_resident + fig_batch_free)."""
        n = int(cbatch.n_gaps)
        cap = int(self.lib.fig_results_capacity(C.byref(self._cm), C.byref(cbatch))) if n > 0 else 1
        fl = np.zeros(max(n, 1), dtype=np.int32); gt = np.zeros(max(n, 1), dtype=np.int32)
        so = np.zeros(n + 1, dtype=np.int64); st = np.zeros(max(cap, 1), dtype=np.uint8)
        r = FigGapResults()
        r.filled_len = _p(fl, c_i32_p); r.gaptofill = _p(gt, c_i32_p); r.str_off = _p(so, c_i64_p)
        r.str = C.cast(st.ctypes.data, C.c_char_p); r.str_capacity = len(st)
        nr = n_ureads + n_preads
        dpos = np.full(max(nr, 1), np.iinfo(np.int32).min, dtype=np.int32); disz = np.zeros(max(nr, 1), dtype=np.int32)
        dlen = np.full(max(2 * n, 1), -1, dtype=np.int32)
        if draw:
            r.draw_pos = _p(dpos, c_i32_p); r.draw_isz = _p(disz, c_i32_p); r.draw_len = _p(dlen, c_i32_p)
        if n > 0 and resident:
            try:
                self._check(self.lib.fig_fill_resident(self.ctx, C.byref(r)), "fig_fill_resident")
            finally:
                self.lib.fig_batch_free(self.ctx)
        elif n > 0:
            self._check(self.lib.fig_fill_gaps(self.ctx, C.byref(cbatch), C.byref(r)), "fig_fill_gaps")
        res = FillResult(fl[:n].copy(), gt[:n].copy(), None, None, str_off=so, raw=st)
        res.draw = (dpos[:nr], disz[:nr], dlen[:2 * n])
        return res

    def fill(self, batch: GapBatch, debug_cand: int = 0, plane_cols: int = 0, plane_reads: int = 0) -> FillResult:
        self.upload(batch)
        try:
            return self.fill_resident(debug_cand, plane_cols, plane_reads)
        finally:
            self.free_batch()

    def stats(self) -> dict:
        s = FigStats()
        self.lib.fig_get_stats(self.ctx, C.byref(s))
        return {k: getattr(s, k) for k, _ in FigStats._fields_}
