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


def _model_pairs(rng, genome: str, contig_idx: int, n_pairs: int, L: int,
                 mu: float, sd: float, err: float, forbid: List[Tuple[int, int]]) -> List[str]:
    """Properly-paired reads for myout.sam (both mates fully inside N-free sequence)."""
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
        r1, _ = _mutate(rng, ref1, err)
        r2, _ = _mutate(rng, ref2, err)
        q = f"m{contig_idx}_{made}"
        lines.append("\t".join([q, "99", str(contig_idx), str(p + 1), f"{L}M", str(isz), r1,
                                _qual(L), _md_tag(ref1, r1), "IH:i:1"]))
        lines.append("\t".join([q, "147", str(contig_idx), str(p + isz - L + 1), f"{L}M",
                                str(-isz), r2, _qual(L), _md_tag(ref2, r2), "IH:i:1"]))
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
              max_reads_per_gap: int = 3000) -> Case:
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
    myout = _model_pairs(rng, truth, 0, n_model_pairs, L, insert_mean, insert_sd, err, forbid)
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
                 gapthresh: int = 400) -> List[str]:
    """The 16 positional args of Figbird.cpp main (Figbird.cpp:6957-6973)."""
    n = len(case.gaps) if n_gaps_for_thread is None else n_gaps_for_thread
    return [paths["scf"], str(case.max_distance), str(case.read_len), str(case.script_itr),
            str(case.partial_flag), str(case.unmapped_flag), str(thread_id), str(n),
            paths["myout"], paths["tmp"], paths["gaps"], str(case.neg_overlap),
            str(case.partial_len), str(gapthresh), "0", str(int(case.insert_mean))]


def fillgaps_argv(case: Case, paths: dict, n_threads: int = 1) -> List[str]:
    """The 15 positional args of FillGaps.cpp main (FillGaps.cpp:419-433)."""
    return [paths["scf"], str(case.max_distance), str(case.read_len), str(case.script_itr),
            str(case.partial_flag), str(case.unmapped_flag), str(n_threads), paths["myout"],
            paths["tmp"], paths["gaps"], str(case.neg_overlap), str(case.partial_len), "10", "0",
            str(int(case.insert_mean))]


def write_gaploads(paths: dict, gap_ids: List[int]) -> None:
    """tmp/gaploads.txt as FillGaps.cpp:313-334 writes it, for a single thread 0."""
    with open(paths["tmp"] + "gaploads.txt", "w") as f:
        f.write("".join(f"{g}\t" for g in gap_ids) + "\n")
