"""Deterministic synthetic inputs at Figbird's post-Preprocess file boundary.

The reference ships no data and bowtie2 is absent, so every test/bench input is
synthesised here, in the formats `Preprocess.cpp` writes and `Figbird.cpp` reads
(SURVEY.md Appendix C):

  scf.fa                     FASTA, header ">0", ">1", ... (numeric = contig index,
                             Figbird.cpp:277-281 `getContigNo` is `atol`)
  tmp/gapInfo.txt            contigIdx \t gapStart0 \t len        (Preprocess.cpp:2144)
  tmp/stat.txt               pairs unmappedPairs maxReadLen 5000  (Preprocess.cpp:2624)
  tmp/stat2.txt              fillflag \t perfectReadGap \t len    (Preprocess.cpp:2628)
  tmp/myout.sam              10-col proper pairs                  (Preprocess.cpp:412-416)
  gaps/gaps_<g>.sam          anchor line + unmapped-mate line     (Preprocess.cpp:404-410)
  gaps/partial_gaps_<g>.sam  7-col soft-clipped reads             (Preprocess.cpp:454,466,478)

Everything derives from one `numpy.random.Generator(PCG64(seed))`, so the GPU box
regenerates identical data from the seed alone.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

import numpy as np

_COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
_ALPH = np.frombuffer(b"ACGT", dtype=np.uint8)


def revcomp(s: str) -> str:
    return "".join(_COMP.get(c, "N") for c in reversed(s))


def _rand_seq(rng, n: int) -> str:
    return _ALPH[rng.integers(0, 4, size=n)].tobytes().decode()


def _mutate(rng, s: str, rate: float) -> Tuple[str, List[int]]:
    """Substitution errors only (0.5 % uniform, no indels)."""
    if rate <= 0:
        return s, []
    b = bytearray(s.encode())
    hits = np.nonzero(rng.random(len(b)) < rate)[0]
    pos = []
    for h in hits:
        old = chr(b[h])
        if old not in "ACGT":
            continue
        new = "ACGT"[(("ACGT".index(old)) + int(rng.integers(1, 4))) % 4]
        b[h] = ord(new)
        pos.append(int(h))
    return b.decode(), pos


def _md_tag(ref: str, read: str) -> str:
    """MD:Z for a gap-free alignment of read vs ref (same length, forward)."""
    out, run = [], 0
    for r, q in zip(ref, read):
        if r == q:
            run += 1
        else:
            out.append(str(run))
            out.append(r)
            run = 0
    out.append(str(run))
    return "MD:Z:" + "".join(out)


@dataclass
class UnmappedRead:
    """One entry of gaps_<g>.sam (two lines): the anchor + its unmapped mate."""
    anchor_pos1: int          # SAM col 4 of the anchor, 1-based
    anchor_reverse: bool      # flag & 16 of the anchor
    anchor_seq: str
    mate_seq_fastq: str       # mate as written to the file (FASTQ orientation)


@dataclass
class PartialRead:
    """One line of partial_gaps_<g>.sam."""
    seq: str
    clipped_index: int
    match: int                # 1/4 left side (fwd/rev), 2/3 right side (fwd/rev)
    pos1: int                 # 1-based
    cigar: str
    mate_pos: int             # or -1
    qual: str


@dataclass
class Gap:
    contig: int
    start: int                # 0-based start of the N run
    length: int               # N-run length in the scaffold (G0)
    truth: str                # true sequence that belongs there ("" for negative overlap)
    stat2: Tuple[int, int, int] = (1, 0, 0)
    unmapped: List[UnmappedRead] = field(default_factory=list)
    partial: List[PartialRead] = field(default_factory=list)


@dataclass
class Case:
    name: str
    mode: str                 # "unmapped" (jump lib) or "partial" (frag lib)
    read_len: int
    insert_mean: float
    insert_sd: float
    max_distance: int         # argv[2] (D)
    partial_len: int          # argv[13]
    neg_overlap: int          # argv[12]
    script_itr: int
    scaffolds: List[str]      # with N runs
    truth: List[str]          # without gaps (may be shorter/longer than the scaffold)
    gaps: List[Gap]
    myout: List[str]          # lines of myout.sam
    n_pairs: int

    @property
    def partial_flag(self) -> int:
        return 1 if self.mode == "partial" else 0

    @property
    def unmapped_flag(self) -> int:
        return 1 if self.mode == "unmapped" else 0


def _qual(n: int) -> str:
    return "I" * n  # flat Q40


def _indel_read(rng, genome: str, p: int, L: int, err: float) -> Tuple[str, str, str]:
    """A read of length L aligned at genome[p:] with ONE insertion or deletion of 1-3 bases away from its ends, plus
    substitution errors: (seq, CIGAR, MD) in Bowtie2's conventions (what processErrorTypes walks, Figbird.cpp:291-487)."""
    k = int(rng.integers(1, 4))
    a = int(rng.integers(8, L - 8 - k))
    if rng.random() < 0.5:                                   # insertion: aMkI(L-a-k)M
        ref = genome[p:p + L - k]
        left, _ = _mutate(rng, ref[:a], err)
        right, _ = _mutate(rng, ref[a:], err)
        seq = left + _rand_seq(rng, k) + right
        cigar = f"{a}M{k}I{L - a - k}M"
        md = _md_tag(ref, left + right)
    else:                                                    # deletion: aMkD(L-a)M, MD "<..>^DEL<..>"
        ref_l, dele, ref_r = genome[p:p + a], genome[p + a:p + a + k], genome[p + a + k:p + L + k]
        left, _ = _mutate(rng, ref_l, err)
        right, _ = _mutate(rng, ref_r, err)
        seq = left + right
        cigar = f"{a}M{k}D{L - a}M"
        md = _md_tag(ref_l, left) + "^" + dele + _md_tag(ref_r, right)[5:]
    return seq, cigar, md


def _model_pairs(rng, genome: str, contig_idx: int, n_pairs: int, L: int,
                 mu: float, sd: float, err: float, forbid: List[Tuple[int, int]], indel_rate: float = 0.0) -> List[str]:
    """Properly-paired reads for myout.sam (both mates fully inside N-free sequence).  `indel_rate` > 0 gives that
    share of the mates an insertion or deletion (CIGAR I/D + MD ^), which feed inPosDist/delPosDist."""
    lines = []
    n = len(genome)
    made = 0
    tries = 0
    while made < n_pairs and tries < n_pairs * 20:
        tries += 1
        isz = int(round(rng.normal(mu, sd)))
        if isz < L + 1 or isz >= n - 2:
            continue
        p = int(rng.integers(0, n - isz))
        bad = False
        for a, b in forbid:
            if p < b + 1 and p + isz > a - 1:
                bad = True
                break
        if bad:
            continue
        ref1 = genome[p:p + L]
        ref2 = genome[p + isz - L:p + isz]
        c1 = c2 = f"{L}M"
        if indel_rate > 0 and rng.random() < indel_rate and p + L + 4 < n:
            r1, c1, md1 = _indel_read(rng, genome, p, L, err)
        else:
            r1, _ = _mutate(rng, ref1, err); md1 = _md_tag(ref1, r1)
        if indel_rate > 0 and rng.random() < indel_rate and p + isz + 4 < n:
            r2, c2, md2 = _indel_read(rng, genome, p + isz - L, L, err)
        else:
            r2, _ = _mutate(rng, ref2, err); md2 = _md_tag(ref2, r2)
        q = f"m{contig_idx}_{made}"
        lines.append("\t".join([q, "99", str(contig_idx), str(p + 1), c1, str(isz), r1,
                                _qual(L), md1, "IH:i:1"]))
        lines.append("\t".join([q, "147", str(contig_idx), str(p + isz - L + 1), c2,
                                str(-isz), r2, _qual(L), md2, "IH:i:1"]))
        made += 1
    return lines


def make_case(name: str, seed: int, mode: str, gap_specs: List[Tuple[int, int]],
              contig_len: int = 12000, read_len: int = 50, insert_mean: float = 600.0,
              insert_sd: float = 30.0, coverage: float = 25.0, err: float = 0.0,
              n_model_pairs: int = 3000, partial_len: Optional[int] = None,
              neg_overlap: int = 30, script_itr: int = 1,
              partial_reads_in_unmapped: bool = True,
              frag_mean: float = 180.0, frag_sd: float = 10.0,
              neg_overlap_gaps: Optional[dict] = None,
              read_n_rate: float = 0.0,
              max_reads_per_gap: int = 3000,
              model_indel_rate: float = 0.0) -> Case:
    """One scaffold with gaps at `gap_specs` = [(start0_in_truth, true_len), ...].

    mode "unmapped": jump library N(insert_mean, insert_sd) -> gaps_<g>.sam (+ frag-library
                     soft-clipped reads in partial_gaps_<g>.sam if requested, as
                     RunFigbird.sh:285 then :338 leave them).
    mode "partial" : frag library N(insert_mean, insert_sd) -> partial_gaps_<g>.sam only.
    `neg_overlap_gaps` maps gap index -> (n_run_len, overlap): the scaffold carries an
    n_run_len N-run although the flanks really overlap by `overlap` bases.
    """
    rng = np.random.default_rng(np.random.PCG64(seed))
    L = read_len
    truth = _rand_seq(rng, contig_len)
    neg_overlap_gaps = neg_overlap_gaps or {}
    # Build the scaffold: replace each truth interval with an N run (same length unless neg-overlap).
    scaf_parts, gaps, cursor, scaf_len = [], [], 0, 0
    truth_to_scaf = []  # (truth_start, truth_end, scaf_start, n_run_len)
    for gi, (ts, tl) in enumerate(sorted(gap_specs)):
        if gi in neg_overlap_gaps:
            nrun, ov = neg_overlap_gaps[gi]
            # left flank = truth[:ts], right flank = truth[ts-ov:], i.e. `ov` bases duplicated.
            scaf_parts.append(truth[cursor:ts])
            scaf_len += ts - cursor
            gaps.append(Gap(0, scaf_len, nrun, ""))
            truth_to_scaf.append((ts, ts - ov, scaf_len, nrun))
            scaf_parts.append("N" * nrun)
            scaf_len += nrun
            cursor = ts - ov
        else:
            scaf_parts.append(truth[cursor:ts])
            scaf_len += ts - cursor
            gaps.append(Gap(0, scaf_len, tl, truth[ts:ts + tl]))
            truth_to_scaf.append((ts, ts + tl, scaf_len, tl))
            scaf_parts.append("N" * tl)
            scaf_len += tl
            cursor = ts + tl
    scaf_parts.append(truth[cursor:])
    scaffold = "".join(scaf_parts)

    def t2s(tpos: int, side: str, gi: int) -> int:
        """truth coordinate -> scaffold coordinate for a base in the flank of gap gi."""
        ts, te, ss, nrun = truth_to_scaf[gi]
        if side == "L":
            return ss - (ts - tpos)
        return ss + nrun + (tpos - te)

    forbid = [(ts - 1, max(te, ts) + 1) for ts, te, _, _ in truth_to_scaf]
    myout = _model_pairs(rng, truth, 0, n_model_pairs, L, insert_mean, insert_sd, err, forbid, model_indel_rate)
    # NB: model pairs carry truth coordinates; positions only feed tlen statistics (col 6),
    # so the small truth/scaffold offset is irrelevant to the model.

    for gi, g in enumerate(gaps):
        ts, te, ss, nrun = truth_to_scaf[gi]
        tl = max(te - ts, 0)
        if mode == "unmapped":
            # number of fragments whose far mate touches the gap ~ coverage * (tl + L) / L per side
            n_side = max(2, int(round(coverage * (tl + L) / (2.0 * L))))
            reads: List[UnmappedRead] = []
            for side in ("L", "R"):
                for _ in range(n_side):
                    isz = int(round(rng.normal(insert_mean, insert_sd)))
                    isz = max(isz, 2 * L + 2)
                    if side == "L":
                        # forward anchor left of gap; mate (reverse strand) overlaps [ts-?, te+?)
                        mate_start = int(rng.integers(ts - L + 8, max(te - 8, ts - L + 9) + 1))
                        frag_end = mate_start + L
                        p0 = frag_end - isz
                        if p0 < 0 or p0 + L > ts:
                            continue
                        if frag_end > len(truth):
                            continue
                        anchor_ref = truth[p0:p0 + L]
                        anchor, _ = _mutate(rng, anchor_ref, err)
                        mate_fwd, _ = _mutate(rng, truth[mate_start:mate_start + L], err)
                        if read_n_rate > 0:
                            mb = bytearray(mate_fwd.encode())
                            for h in np.nonzero(rng.random(L) < read_n_rate)[0]:
                                mb[h] = ord("N")
                            mate_fwd = mb.decode()
                        reads.append(UnmappedRead(t2s(p0, "L", gi) + 1, False, anchor,
                                                  revcomp(mate_fwd)))
                    else:
                        mate_start = int(rng.integers(ts - L + 8, max(te - 8, ts - L + 9) + 1))
                        q0 = mate_start + isz - L  # anchor start (reverse strand), right of gap
                        if q0 < te or q0 + L > len(truth) or mate_start < 0:
                            continue
                        anchor_ref = truth[q0:q0 + L]
                        anchor, _ = _mutate(rng, anchor_ref, err)
                        mate_fwd, _ = _mutate(rng, truth[mate_start:mate_start + L], err)
                        reads.append(UnmappedRead(t2s(q0, "R", gi) + 1, True, anchor, mate_fwd))
            order = rng.permutation(len(reads))
            g.unmapped = [reads[i] for i in order][:max_reads_per_gap]
        want_partial = (mode == "partial") or partial_reads_in_unmapped
        if want_partial:
            fm, fs = (insert_mean, insert_sd) if mode == "partial" else (frag_mean, frag_sd)
            n_side = max(2, int(round(coverage * 0.6)))
            preads: List[PartialRead] = []
            for side in ("L", "R"):
                for _ in range(n_side):
                    fwd = bool(rng.integers(0, 2))
                    if side == "L":
                        # aligned part left of gap, tail clipped into the gap
                        aligned = int(rng.integers(6, L - 5))
                        s = ts - aligned          # truth start of read
                        if s < 0 or s + L > len(truth):
                            continue
                        seq, _ = _mutate(rng, truth[s:s + L], err)
                        pos1 = t2s(s, "L", gi) + 1
                        clipped_index = g.start - pos1
                        cigar = f"{aligned}M{L - aligned}S"
                        isz = int(round(rng.normal(fm, fs)))
                        mate = pos1 - (isz - L) if fwd is False else pos1 + (isz - L)
                        mate_pos = mate if (rng.random() < 0.8 and mate > 0) else -1
                        # left-side reads anchor on the left => their mate lies further left
                        if mate_pos != -1:
                            mate_pos = max(1, pos1 - (isz - L))
                        preads.append(PartialRead(seq, clipped_index, 1 if fwd else 4, pos1, cigar,
                                                  mate_pos, _qual(L)))
                    else:
                        aligned = int(rng.integers(6, L - 5))
                        clip = L - aligned
                        s = te - clip             # truth start of read (clip bases inside gap)
                        if s < 0 or s + L > len(truth):
                            continue
                        if clip > tl and gi not in neg_overlap_gaps:
                            # read would run past the gap into the left flank: still fine, it is
                            # how a short gap looks; keep it.
                            pass
                        seq, _ = _mutate(rng, truth[s:s + L], err)
                        pos1 = g.start + g.length + 1   # first aligned base after the gap, 1-based
                        clipped_index = clip
                        cigar = f"{clip}S{aligned}M"
                        isz = int(round(rng.normal(fm, fs)))
                        mate_pos = pos1 + (isz - L) if rng.random() < 0.8 else -1
                        preads.append(PartialRead(seq, clipped_index, 2 if fwd else 3, pos1, cigar,
                                                  mate_pos, _qual(L)))
            order = rng.permutation(len(preads))
            g.partial = [preads[i] for i in order][:max_reads_per_gap]

    D = int(insert_mean) if mode == "partial" else int(1.15 * insert_mean)
    return Case(name=name, mode=mode, read_len=L, insert_mean=insert_mean, insert_sd=insert_sd,
                max_distance=D, partial_len=partial_len or L, neg_overlap=neg_overlap,
                script_itr=script_itr, scaffolds=[scaffold], truth=[truth], gaps=gaps,
                myout=myout, n_pairs=len(myout) // 2)


def write_case(case: Case, root: str) -> dict:
    """Write the file layout Figbird.cpp / FillGaps.cpp expect; returns the path map."""
    tmp = os.path.join(root, "tmp") + "/"
    gdir = os.path.join(root, "gaps") + "/"
    os.makedirs(tmp, exist_ok=True)
    os.makedirs(gdir, exist_ok=True)
    scf = os.path.join(root, "scf.fa")
    with open(scf, "w") as f:
        for i, s in enumerate(case.scaffolds):
            f.write(f">{i}\n")
            for k in range(0, len(s), 60):
                f.write(s[k:k + 60] + "\n")
    with open(tmp + "gapInfo.txt", "w") as f:
        for g in case.gaps:
            f.write(f"{g.contig}\t{g.start}\t{g.length}\n")
    with open(tmp + "stat.txt", "w") as f:
        f.write(f"{case.n_pairs} 0 {case.read_len} 5000")
    with open(tmp + "stat2.txt", "w") as f:
        for g in case.gaps:
            f.write("\t".join(str(x) for x in g.stat2) + "\n")
    with open(tmp + "myout.sam", "w") as f:
        for ln in case.myout:
            f.write(ln + "\n")
    L = case.read_len
    for gi, g in enumerate(case.gaps):
        with open(gdir + f"gaps_{gi}.sam", "w") as f:
            for k, r in enumerate(g.unmapped):
                q = f"u{gi}_{k}"
                fl1, fl2 = (89, 165) if r.anchor_reverse else (73, 133)
                f.write("\t".join([q, str(fl1), str(g.contig), str(r.anchor_pos1), f"{L}M", "0",
                                   r.anchor_seq, _qual(len(r.anchor_seq)), f"MD:Z:{L}",
                                   "IH:i:1"]) + "\n")
                f.write("\t".join([q, str(fl2), str(g.contig), str(r.anchor_pos1), "*", "0",
                                   r.mate_seq_fastq, _qual(len(r.mate_seq_fastq)), f"MD:Z:{L}",
                                   "IH:i:1"]) + "\n")
        with open(gdir + f"partial_gaps_{gi}.sam", "w") as f:
            for r in g.partial:
                f.write("\t".join([r.seq, str(r.clipped_index), str(r.match), str(r.pos1), r.cigar,
                                   str(r.mate_pos), r.qual]) + "\n")
    return {"scf": scf, "tmp": tmp, "gaps": gdir, "myout": tmp + "myout.sam"}


def figbird_argv(case: Case, paths: dict, thread_id: int = 0, n_gaps_for_thread: Optional[int] = None,
                 gapthresh: int = 400, set_inputmean: int = 0) -> List[str]:
    """The 16 positional args of Figbird.cpp main (Figbird.cpp:6957-6973)."""
    n = len(case.gaps) if n_gaps_for_thread is None else n_gaps_for_thread
    return [paths["scf"], str(case.max_distance), str(case.read_len), str(case.script_itr),
            str(case.partial_flag), str(case.unmapped_flag), str(thread_id), str(n),
            paths["myout"], paths["tmp"], paths["gaps"], str(case.neg_overlap),
            str(case.partial_len), str(gapthresh), str(set_inputmean), str(int(case.insert_mean))]


def fillgaps_argv(case: Case, paths: dict, n_threads: int = 1, set_inputmean: int = 0) -> List[str]:
    """The 15 positional args of FillGaps.cpp main (FillGaps.cpp:419-433)."""
    return [paths["scf"], str(case.max_distance), str(case.read_len), str(case.script_itr),
            str(case.partial_flag), str(case.unmapped_flag), str(n_threads), paths["myout"],
            paths["tmp"], paths["gaps"], str(case.neg_overlap), str(case.partial_len), "10", str(set_inputmean),
            str(int(case.insert_mean))]


def write_gaploads(paths: dict, gap_ids: List[int]) -> None:
    """tmp/gaploads.txt as FillGaps.cpp:313-334 writes it, for a single thread 0."""
    with open(paths["tmp"] + "gaploads.txt", "w") as f:
        f.write("".join(f"{g}\t" for g in gap_ids) + "\n")


# =========================================================================================
# In-memory batches for the C ABI (figbird_amd.api.GapBatch) -- used by tests and bench.py
# =========================================================================================
def case_to_batch(case: Case):
    """Case -> GapBatch, applying what parseUnmapped does to each mate (Figbird.cpp:5727-5736):
    reverse-complement it iff the anchor is on the forward strand."""
    from .api import GapBatch
    seq = "".join(s.upper() for s in case.scaffolds)
    off = np.zeros(len(case.scaffolds) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(s) for s in case.scaffolds])
    u_off, u_pos, u_rev, u_soff, u_seq = [0], [], [], [0], []
    p_off, p_clip, p_match, p_pos, p_ref, p_soff, p_seq, p_qual = [0], [], [], [], [], [0], [], []
    fillflag = []
    for g in case.gaps:
        n = 0
        if case.mode == "unmapped":
            for r in g.unmapped[:3000]:
                s = r.mate_seq_fastq if r.anchor_reverse else revcomp(r.mate_seq_fastq)
                u_pos.append(r.anchor_pos1); u_rev.append(0 if r.anchor_reverse else 1)
                u_seq.append(s); u_soff.append(u_soff[-1] + len(s)); n += 1
        u_off.append(u_off[-1] + n)
        fillflag.append(-1 if (case.mode == "unmapped" and len(g.unmapped) > 3000) else 1)
        for r in g.partial[:3001]:
            p_clip.append(r.clipped_index); p_match.append(r.match); p_pos.append(r.pos1); p_ref.append(r.mate_pos)
            p_seq.append(r.seq); p_qual.append(r.qual); p_soff.append(p_soff[-1] + len(r.seq))
        p_off.append(p_off[-1] + min(len(g.partial), 3001))
    a8 = lambda s: np.frombuffer((s or "\0").encode(), dtype=np.uint8).copy()
    i32 = lambda v: np.asarray(v if len(v) else [0], dtype=np.int32)
    return GapBatch(
        contig_off=off, contig_seq=a8(seq),
        gap_contig=i32([g.contig for g in case.gaps]), gap_start=np.asarray([g.start for g in case.gaps], dtype=np.int64),
        gap_len=i32([g.length for g in case.gaps]), gap_stat2=i32([x for g in case.gaps for x in g.stat2]),
        gap_fillflag=i32(fillflag),
        u_read_off=np.asarray(u_off, dtype=np.int64), u_anchor_pos=i32(u_pos), u_is_reverse=np.asarray(u_rev if u_rev else [0], dtype=np.uint8),
        u_seq_off=np.asarray(u_soff, dtype=np.int64), u_seq=a8("".join(u_seq)),
        p_read_off=np.asarray(p_off, dtype=np.int64), p_clipped_index=i32(p_clip), p_match=i32(p_match), p_pos=i32(p_pos),
        p_ref_pos=i32(p_ref), p_seq_off=np.asarray(p_soff, dtype=np.int64), p_seq=a8("".join(p_seq)), p_qual=a8("".join(p_qual)))


_COMP_LUT = np.full(256, ord("N"), dtype=np.uint8)
for _a, _b in zip(b"ACGT", b"TGCA"):
    _COMP_LUT[_a] = _b


@dataclass
class BenchSpec:
    """The synthetic set of BASELINE.json / SURVEY.md §8(d): scaffolds of 50 kb, a gap every
    ~5 kb, two libraries (frag 2x101 N(180,10) -> partial mode, jump 2x150 N(3500,350) ->
    unmapped mode), 0.5 % substitutions, flat Q40, reads/gap ~ coverage-proportional with the
    reference's 3000-read cap."""
    mode: str = "unmapped"
    read_len: int = 150
    insert_mean: float = 3500.0
    insert_sd: float = 350.0
    err: float = 0.005
    reads_per_gap_mean: float = 1000.0      # 1e8 reads / 1e5 gaps
    gap_mix: str = "gage"                   # "gage" (<=30:20%,31-133:30%,134-400:25%,>400:25%) or "loguniform" [50,2000]
    scaffold_len: int = 50000
    gap_spacing: int = 5000
    partial_cov: int = 24
    frag_len: int = 101

    @property
    def max_distance(self) -> int:
        return int(self.insert_mean) if self.mode == "partial" else int(1.15 * self.insert_mean)


def _draw_gap_lengths(rng, n: int, mix: str) -> np.ndarray:
    if mix == "loguniform":
        return np.exp(rng.uniform(np.log(50), np.log(2000), size=n)).astype(np.int64)
    u = rng.random(n)
    out = np.empty(n, dtype=np.int64)
    a = u < 0.20; b = (u >= 0.20) & (u < 0.50); c = (u >= 0.50) & (u < 0.75); d = u >= 0.75
    out[a] = rng.integers(5, 31, size=int(a.sum()))
    out[b] = rng.integers(31, 134, size=int(b.sum()))
    out[c] = rng.integers(134, 401, size=int(c.sum()))
    out[d] = np.exp(rng.uniform(np.log(401), np.log(2000), size=int(d.sum()))).astype(np.int64)
    return out


def make_bench_batch(seed: int, n_gaps: int, spec: BenchSpec, gap_lengths: Optional[np.ndarray] = None):
    """Vectorised generator: returns (GapBatch, truth) with `truth[g]` the true gap sequence (uint8)."""
    from .api import GapBatch
    rng = np.random.default_rng(np.random.PCG64(seed))
    L = spec.read_len
    G = _draw_gap_lengths(rng, n_gaps, spec.gap_mix) if gap_lengths is None else np.asarray(gap_lengths, dtype=np.int64)
    per_scaf = max(1, spec.scaffold_len // spec.gap_spacing - 1)
    n_scaf = (n_gaps + per_scaf - 1) // per_scaf
    contig_seqs, contig_off = [], [0]
    gap_contig = np.zeros(n_gaps, dtype=np.int32); gap_start = np.zeros(n_gaps, dtype=np.int64)
    truths = []
    truth_scaf = []                                  # per scaffold: truth array and (gap truth start) list
    gi = 0
    for s in range(n_scaf):
        t = _ALPH[rng.integers(0, 4, size=spec.scaffold_len)]
        scaf = t.copy()
        starts = []
        for k in range(per_scaf):
            if gi >= n_gaps:
                break
            st = (k + 1) * spec.gap_spacing - int(G[gi]) // 2
            scaf[st:st + G[gi]] = ord("N")
            gap_contig[gi] = s; gap_start[gi] = st
            truths.append(t[st:st + G[gi]].copy())
            starts.append((gi, st))
            gi += 1
        contig_seqs.append(scaf); contig_off.append(contig_off[-1] + len(scaf))
        truth_scaf.append((t, starts))
    mu, sd = spec.insert_mean, spec.insert_sd
    u_off, u_pos, u_rev, u_seqs = [0], [], [], []
    p_off, p_clip, p_match, p_pos, p_ref, p_seqs = [0], [], [], [], [], []
    meanG = float(np.mean(G + L))
    for s, (t, starts) in enumerate(truth_scaf):
        for g, st in starts:
            g0 = int(G[g]); te = st + g0
            nU = 0
            if spec.mode == "unmapped":
                n = int(round(spec.reads_per_gap_mean * (g0 + L) / meanG))
                n = max(4, min(n, 3000))
                left = rng.random(n) < 0.5
                lo, hi = st - L + 8, max(te - 8, st - L + 9)
                ms = rng.integers(lo, hi + 1, size=n)
                isz = np.maximum(np.rint(rng.normal(mu, sd, size=n)).astype(np.int64), 2 * L + 2)
                # left: forward anchor at p0 = ms + L - isz ; right: reverse anchor at q0 = ms + isz - L
                p0 = ms + L - isz; q0 = ms + isz - L
                ok = np.where(left, (p0 >= 0) & (p0 + L <= st), (q0 >= te) & (q0 + L <= len(t)))
                ms, left, p0, q0 = ms[ok], left[ok], p0[ok], q0[ok]
                n = len(ms)
                reads = t[ms[:, None] + np.arange(L)[None, :]]
                if spec.err > 0:
                    hit = rng.random(reads.shape) < spec.err
                    sub = _ALPH[(np.searchsorted(_ALPH, reads) + rng.integers(1, 4, size=reads.shape)) % 4]
                    reads = np.where(hit, sub, reads)
                u_pos.append(np.where(left, p0 + 1, q0 + 1).astype(np.int32))
                u_rev.append(left.astype(np.uint8))
                u_seqs.append(reads.reshape(-1))
                nU = n
            u_off.append(u_off[-1] + nU)
            # frag-library soft-clipped reads (both modes, RunFigbird.sh:285 then :338)
            Lp = spec.frag_len if spec.mode == "unmapped" else L
            npart = spec.partial_cov
            aligned = rng.integers(6, Lp - 5, size=npart)
            is_left = rng.random(npart) < 0.5
            fwd = rng.random(npart) < 0.5
            rs = np.where(is_left, st - aligned, te - (Lp - aligned))
            okp = (rs >= 0) & (rs + Lp <= len(t))
            aligned, is_left, fwd, rs = aligned[okp], is_left[okp], fwd[okp], rs[okp]
            preads = t[rs[:, None] + np.arange(Lp)[None, :]]
            if spec.err > 0 and len(rs):
                hit = rng.random(preads.shape) < spec.err
                sub = _ALPH[(np.searchsorted(_ALPH, preads) + rng.integers(1, 4, size=preads.shape)) % 4]
                preads = np.where(hit, sub, preads)
            pos1 = np.where(is_left, rs + 1, te + 1)
            clip = np.where(is_left, st - pos1, Lp - aligned)
            fm, fs = (180.0, 10.0) if spec.mode == "unmapped" else (mu, sd)
            fisz = np.rint(rng.normal(fm, fs, size=len(rs))).astype(np.int64)
            has_mate = rng.random(len(rs)) < 0.8
            mate = np.where(is_left, np.maximum(1, pos1 - (fisz - Lp)), pos1 + (fisz - Lp))
            mate = np.where(has_mate, mate, -1)
            p_clip.append(clip.astype(np.int32)); p_pos.append(pos1.astype(np.int32)); p_ref.append(mate.astype(np.int32))
            p_match.append(np.where(is_left, np.where(fwd, 1, 4), np.where(fwd, 2, 3)).astype(np.int32))
            p_seqs.append(preads.reshape(-1)); p_off.append(p_off[-1] + len(rs))
    cat = lambda lst, dt: (np.concatenate(lst).astype(dt) if lst and sum(len(x) for x in lst) else np.zeros(1, dtype=dt))
    u_seq = cat(u_seqs, np.uint8); p_seq = cat(p_seqs, np.uint8)
    nu = u_off[-1]; npr = p_off[-1]
    Lp = spec.frag_len if spec.mode == "unmapped" else L
    batch = GapBatch(
        contig_off=np.asarray(contig_off, dtype=np.int64), contig_seq=np.concatenate(contig_seqs),
        gap_contig=gap_contig, gap_start=gap_start, gap_len=G.astype(np.int32),
        gap_stat2=np.tile(np.asarray([1, 0, 0], dtype=np.int32), n_gaps), gap_fillflag=np.ones(n_gaps, dtype=np.int32),
        u_read_off=np.asarray(u_off, dtype=np.int64), u_anchor_pos=cat(u_pos, np.int32), u_is_reverse=cat(u_rev, np.uint8),
        u_seq_off=(np.arange(nu + 1, dtype=np.int64) * L), u_seq=u_seq,
        p_read_off=np.asarray(p_off, dtype=np.int64), p_clipped_index=cat(p_clip, np.int32), p_match=cat(p_match, np.int32),
        p_pos=cat(p_pos, np.int32), p_ref_pos=cat(p_ref, np.int32), p_seq_off=(np.arange(npr + 1, dtype=np.int64) * Lp),
        p_seq=p_seq, p_qual=np.full(max(len(p_seq), 1), ord("I"), dtype=np.uint8))
    return batch, truths


def bench_model_case(seed: int, spec: BenchSpec, n_pairs: int = 4000) -> Case:
    """A small gap-free Case whose myout.sam / stat.txt define the run-level model of a bench run."""
    c = make_case(f"bench_model_{spec.mode}", seed, spec.mode, [(2000, 10)], contig_len=4 * int(spec.insert_mean) + 6000,
                  read_len=spec.read_len, insert_mean=spec.insert_mean, insert_sd=spec.insert_sd, coverage=1, err=spec.err,
                  n_model_pairs=n_pairs, partial_len=(spec.frag_len if spec.mode == "unmapped" else spec.read_len))
    return c


def write_batch_subset(batch, gap_ids, model_case: Case, root: str, spec: BenchSpec) -> dict:
    """Write the post-Preprocess file layout for a subset of a GapBatch (CPU-baseline sample):
    the selected gaps are renumbered 0..k-1; their scaffolds are written whole."""
    tmp = os.path.join(root, "tmp") + "/"; gdir = os.path.join(root, "gaps") + "/"
    os.makedirs(tmp, exist_ok=True); os.makedirs(gdir, exist_ok=True)
    contigs = sorted(set(int(batch.gap_contig[g]) for g in gap_ids))
    remap = {c: i for i, c in enumerate(contigs)}
    scf = os.path.join(root, "scf.fa")
    with open(scf, "w") as f:
        for c in contigs:
            s = batch.contig_seq[batch.contig_off[c]:batch.contig_off[c + 1]].tobytes().decode()
            f.write(f">{remap[c]}\n")
            for k in range(0, len(s), 60):
                f.write(s[k:k + 60] + "\n")
    # Every N run of a written scaffold must have a gapInfo line (FillGaps.cpp pairs them in file order),
    # so all gaps of the selected scaffolds are listed; unselected ones get fillflag via empty read files.
    all_g = [g for g in range(batch.n_gaps) if int(batch.gap_contig[g]) in remap]
    all_g.sort(key=lambda g: (remap[int(batch.gap_contig[g])], int(batch.gap_start[g])))
    sel = set(int(g) for g in gap_ids)
    with open(tmp + "gapInfo.txt", "w") as f1, open(tmp + "stat2.txt", "w") as f2:
        for g in all_g:
            f1.write(f"{remap[int(batch.gap_contig[g])]}\t{int(batch.gap_start[g])}\t{int(batch.gap_len[g])}\n")
            f2.write("1\t0\t0\n")
    with open(tmp + "stat.txt", "w") as f:
        f.write(f"{model_case.n_pairs} 0 {model_case.read_len} 5000")
    with open(tmp + "myout.sam", "w") as f:
        for ln in model_case.myout:
            # model pairs live on contig 0 of the model case; point them at contig 0 here (only its length matters)
            f.write(ln + "\n")
    L = spec.read_len
    anchor = "A" * L
    for k, g in enumerate(all_g):
        with open(gdir + f"gaps_{k}.sam", "w") as f:
            if g in sel and spec.mode == "unmapped":
                for r in range(int(batch.u_read_off[g]), int(batch.u_read_off[g + 1])):
                    s = batch.u_seq[batch.u_seq_off[r]:batch.u_seq_off[r + 1]]
                    rev = int(batch.u_is_reverse[r])
                    mate = (_COMP_LUT[s][::-1] if rev else s).tobytes().decode()
                    fl1, fl2 = (73, 133) if rev else (89, 165)
                    q = f"u{k}_{r}"
                    f.write("\t".join([q, str(fl1), "0", str(int(batch.u_anchor_pos[r])), f"{L}M", "0", anchor, _qual(L), f"MD:Z:{L}", "IH:i:1"]) + "\n")
                    f.write("\t".join([q, str(fl2), "0", str(int(batch.u_anchor_pos[r])), "*", "0", mate, _qual(len(mate)), f"MD:Z:{L}", "IH:i:1"]) + "\n")
        with open(gdir + f"partial_gaps_{k}.sam", "w") as f:
            if g in sel:
                for r in range(int(batch.p_read_off[g]), int(batch.p_read_off[g + 1])):
                    s = batch.p_seq[batch.p_seq_off[r]:batch.p_seq_off[r + 1]].tobytes().decode()
                    f.write("\t".join([s, str(int(batch.p_clipped_index[r])), str(int(batch.p_match[r])), str(int(batch.p_pos[r])),
                                       "*", str(int(batch.p_ref_pos[r])), _qual(len(s))]) + "\n")
    return {"scf": scf, "tmp": tmp, "gaps": gdir, "myout": tmp + "myout.sam", "gap_order": all_g}


def subset_batch(batch, gap_ids):
    """A GapBatch holding only `gap_ids` (in the given order); scaffolds are kept whole (they are shared)."""
    from .api import GapBatch
    gap_ids = [int(g) for g in gap_ids]

    def take(off, *arrs):
        idx = np.concatenate([np.arange(off[g], off[g + 1]) for g in gap_ids]) if gap_ids else np.zeros(0, dtype=np.int64)
        new_off = np.zeros(len(gap_ids) + 1, dtype=np.int64)
        new_off[1:] = np.cumsum([off[g + 1] - off[g] for g in gap_ids])
        return idx.astype(np.int64), new_off, [a[idx] if len(idx) else a[:0] for a in arrs]

    def take_seq(seq_off, seq, ridx):
        lens = (seq_off[1:] - seq_off[:-1])[ridx] if len(ridx) else np.zeros(0, dtype=np.int64)
        new_off = np.zeros(len(ridx) + 1, dtype=np.int64)
        new_off[1:] = np.cumsum(lens)
        if len(ridx):
            pieces = [seq[seq_off[r]:seq_off[r + 1]] for r in ridx]
            new_seq = np.concatenate(pieces)
        else:
            new_seq = np.zeros(1, dtype=np.uint8)
        return new_off, new_seq

    uidx, u_off, (u_pos, u_rev) = take(batch.u_read_off, batch.u_anchor_pos, batch.u_is_reverse)
    pidx, p_off, (p_clip, p_match, p_pos, p_ref) = take(batch.p_read_off, batch.p_clipped_index, batch.p_match, batch.p_pos, batch.p_ref_pos)
    u_soff, u_seq = take_seq(batch.u_seq_off, batch.u_seq, uidx)
    p_soff, p_seq = take_seq(batch.p_seq_off, batch.p_seq, pidx)
    _, p_qual = take_seq(batch.p_seq_off, batch.p_qual, pidx)
    pad = lambda a, dt: (a.astype(dt) if len(a) else np.zeros(1, dtype=dt))
    gi = np.asarray(gap_ids, dtype=np.int64)
    return GapBatch(contig_off=batch.contig_off, contig_seq=batch.contig_seq,
                    gap_contig=pad(batch.gap_contig[gi], np.int32), gap_start=pad(batch.gap_start[gi], np.int64), gap_len=pad(batch.gap_len[gi], np.int32),
                    gap_stat2=pad(batch.gap_stat2.reshape(-1, 3)[gi].reshape(-1), np.int32), gap_fillflag=pad(batch.gap_fillflag[gi], np.int32),
                    u_read_off=u_off, u_anchor_pos=pad(u_pos, np.int32), u_is_reverse=pad(u_rev, np.uint8), u_seq_off=u_soff, u_seq=u_seq,
                    p_read_off=p_off, p_clipped_index=pad(p_clip, np.int32), p_match=pad(p_match, np.int32), p_pos=pad(p_pos, np.int32),
                    p_ref_pos=pad(p_ref, np.int32), p_seq_off=p_soff, p_seq=p_seq, p_qual=p_qual)
