#!/usr/bin/env python3
"""Synthetic Bowtie2-style SAM at the pre-Preprocess boundary: what `bowtie2 --local -X maxD` (frag
library, samflag 1) and `bowtie2 -X isz*1.15` (jump library, samflag 2) hand to Preprocess.cpp.  bowtie2 is absent from
the container, so the aligner is stubbed: reads are cut from a known truth and given the records an aligner would write
(11 mandatory columns + AS/XN/XM/XO/XG/NM/YS/YT/MD tags; mates adjacent; unaligned mates without MD/NM/AS).  The mix is
deliberately broader than real data (mirror-unmapped pairs, both-unmapped pairs, N-rich reads, duplicates, xMyIzM reads
over short gaps, improper pairs) so that every branch of the binning runs; the reference's own binary is the judge."""
from __future__ import annotations

import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from figbird_amd import synth  # noqa: E402


def make_scaffolds(rng, n_contigs=2, contig_len=6000, gaps_per_contig=3, end_gap=False, gapless_first=False):
    """-> (truths, scaffolds, gaps[(contig, start, nrun_len, true_len)])"""
    truths, scafs, gaps = [], [], []
    for c in range(n_contigs):
        t = synth._rand_seq(rng, contig_len)
        cuts = sorted(int(x) for x in rng.choice(np.arange(700, contig_len - 900, 40), size=gaps_per_contig, replace=False))
        if gapless_first and c == 0 and n_contigs > 1:
            cuts = []
        parts, cur = [], 0
        scaf_pos = 0
        for k, st in enumerate(cuts):
            if st < cur + 300:
                continue
            tl = int(rng.choice([3, 12, 25, 60, 140, 320]))
            nl = tl if rng.random() < 0.7 else max(1, tl + int(rng.integers(-2, 6)))
            parts.append(t[cur:st]); scaf_pos += st - cur
            gaps.append((c, scaf_pos, nl, tl, st))
            parts.append("N" * nl); scaf_pos += nl
            cur = st + tl
        parts.append(t[cur:])
        s = "".join(parts)
        if end_gap and c == 0:
            s = s + "N" * 17                      # an N-run that reaches the contig's end (the carried-over gap quirk)
        truths.append(t); scafs.append(s)
    return truths, scafs, gaps


def _tags(ref, read, extra=""):
    nm = sum(1 for a, b in zip(ref, read) if a != b)
    return f"AS:i:{-6 * nm}\tXN:i:0\tXM:i:{nm}\tXO:i:0\tXG:i:0\tNM:i:{nm}\tYS:i:0\tYT:Z:CP\t{synth._md_tag(ref, read)}{extra}"


def make_sam(seed, truths, scafs, gaps, L, isz_mean, isz_sd, n_pairs, local_mode, names=None, err=0.004, dup_rate=0.03, n_rate=0.02, md_on_unaligned=False):
    """One library's SAM text.  Coordinates are scaffold coordinates; reads come from the truth and are placed on the
    scaffold through the nearest gap's offset."""
    rng = np.random.default_rng(np.random.PCG64(seed))
    names = names or [f"scf{c}" for c in range(len(scafs))]
    out = ["@HD\tVN:1.0\tSO:unsorted"] + [f"@SQ\tSN:{names[c]}\tLN:{len(scafs[c])}" for c in range(len(scafs))] + ["@PG\tID:bowtie2\tPN:bowtie2\tVN:2.2.3"]
    by_c = {}
    for g in gaps:
        by_c.setdefault(g[0], []).append(g)

    def to_scaf(c, tpos):
        """truth position -> (scaffold position, inside_gap?)"""
        off = 0
        for (_, sstart, nl, tl, tstart) in by_c.get(c, []):
            if tpos < tstart:
                break
            if tpos < tstart + tl:
                return sstart, True
            off += nl - tl
        return tpos + off, False

    def mate_record(c, tstart, rev):
        """aligner's view of a read cut at truth[tstart:tstart+L]: ('aligned', pos1, cigar, seq, ref) | ('unaligned', seq)"""
        t = truths[c]
        raw = t[tstart:tstart + L]
        seq, _ = synth._mutate(rng, raw, err)
        if rng.random() < n_rate:
            b = bytearray(seq.encode())
            for h in rng.integers(0, L, size=int(rng.integers(1, 6))):
                b[h] = ord("N")
            seq = b.decode()
        # bases of the read that lie inside a gap of the truth
        in_gap = [to_scaf(c, tstart + k)[1] for k in range(L)]
        if not any(in_gap):
            p, _ = to_scaf(c, tstart)
            return ("aligned", p + 1, f"{L}M", seq, raw, seq)
        k0 = in_gap.index(True); k1 = L - 1 - in_gap[::-1].index(True)
        left_al, right_al = k0, L - 1 - k1
        # a read across a short gap whose N-run is shorter than the truth: an insertion relative to the scaffold (xMyIzM)
        gi = [g for g in by_c[c] if g[4] <= tstart + k0 < g[4] + g[3]][0]
        if left_al >= 12 and right_al >= 12 and gi[3] <= 20 and rng.random() < 0.5:
            p, _ = to_scaf(c, tstart)
            return ("aligned", p + 1, f"{left_al}M{k1 - k0 + 1}I{right_al}M", seq, raw[:left_al] + raw[k1 + 1:], seq[:left_al] + seq[k1 + 1:])
        if not local_mode or max(left_al, right_al) < 20:
            return ("unaligned", seq)
        if left_al >= right_al:
            p, _ = to_scaf(c, tstart)
            return ("aligned", p + 1, f"{left_al}M{L - left_al}S", seq, raw[:left_al], seq[:left_al])
        p, _ = to_scaf(c, tstart + k1 + 1)
        cig = f"{L - right_al}S{right_al}M"
        if rng.random() < 0.1 and right_al > 30:
            return ("aligned", p + 1, f"{L - right_al}S{right_al - 8}M8S", seq, raw[k1 + 1:L - 8], seq[k1 + 1:L - 8])     # S..M..S
        return ("aligned", p + 1, cig, seq, raw[k1 + 1:], seq[k1 + 1:])

    prev = None
    for n in range(n_pairs):
        if prev is not None and rng.random() < dup_rate:
            c, p0, isz = prev
        else:
            c = int(rng.integers(0, len(truths)))
            isz = int(round(rng.normal(isz_mean, isz_sd)))
            isz = max(isz, L + 5)
            if isz >= len(truths[c]) - 2:
                continue
            p0 = int(rng.integers(0, len(truths[c]) - isz))
            prev = (c, p0, isz)
        q = f"r{seed}_{n}"
        a = mate_record(c, p0, False)
        b = mate_record(c, p0 + isz - L, True)
        qual = "I" * L
        swap = rng.random() < 0.5                                    # which mate is /1
        first, second = (a, b) if not swap else (b, a)
        first_rev = swap; second_rev = not swap
        kind = rng.random()
        if kind < 0.015:                                             # a pair the aligner could not place at all
            out.append("\t".join([q, "77", "*", "0", "0", "*", "*", "0", "0", first[3] if first[0] == "aligned" else first[1], qual, "YT:Z:UP"]))
            out.append("\t".join([q, "141", "*", "0", "0", "*", "*", "0", "0", second[3] if second[0] == "aligned" else second[1], qual, "YT:Z:UP"]))
            continue
        if first[0] == "aligned" and second[0] == "aligned":
            improper = kind > 0.97
            f1 = (1 | (0 if improper else 2) | (16 if first_rev else 0) | (32 if second_rev else 0) | 64)
            f2 = (1 | (0 if improper else 2) | (16 if second_rev else 0) | (32 if first_rev else 0) | 128)
            tl = isz if not first_rev else -isz
            out.append("\t".join([q, str(f1), names[c], str(first[1]), "42", first[2], "=", str(second[1]), str(tl), first[3], qual, _tags(first[4], first[5])]))
            out.append("\t".join([q, str(f2), names[c], str(second[1]), "42", second[2], "=", str(first[1]), str(-tl), second[3], qual, _tags(second[4], second[5])]))
        elif first[0] == "aligned":                                   # mate 1 placed, mate 2 not: the pairs Figbird bins to gaps
            f1 = 1 | 8 | (16 if first_rev else 0) | 64
            f2 = 1 | 4 | (32 if first_rev else 0) | 128
            mate_seq = second[1] if not second_rev else synth.revcomp(second[1])             # unaligned mates keep FASTQ orientation
            extra = "\tMD:Z:0" if md_on_unaligned else ""          # (a tag bowtie2 does not write; it pins the reference's otherwise undefined column)
            out.append("\t".join([q, str(f1), names[c], str(first[1]), "42", first[2], "=", str(first[1]), "0", first[3], qual, _tags(first[4], first[5]).replace("YT:Z:CP", "YT:Z:UP")]))
            out.append("\t".join([q, str(f2), names[c], str(first[1]), "0", "*", "=", str(first[1]), "0", mate_seq, qual, "YS:i:0\tYT:Z:UP" + extra]))
        elif second[0] == "aligned":                                  # the mirror case (commented out in the reference)
            f1 = 1 | 4 | (32 if second_rev else 0) | 64
            f2 = 1 | 8 | (16 if second_rev else 0) | 128
            mate_seq = first[1] if not first_rev else synth.revcomp(first[1])
            out.append("\t".join([q, str(f1), names[c], str(second[1]), "0", "*", "=", str(second[1]), "0", mate_seq, qual, "YS:i:0\tYT:Z:UP"]))
            out.append("\t".join([q, str(f2), names[c], str(second[1]), "42", second[2], "=", str(second[1]), "0", second[3], qual, _tags(second[4], second[5]).replace("YT:Z:CP", "YT:Z:UP")]))
        else:
            out.append("\t".join([q, "77", "*", "0", "0", "*", "*", "0", "0", first[1], qual, "YT:Z:UP"]))
            out.append("\t".join([q, "141", "*", "0", "0", "*", "*", "0", "0", second[1], qual, "YT:Z:UP"]))
    return "\n".join(out) + "\n"


def write_fasta(path, names, scafs, width=60):
    with open(path, "w") as f:
        for n, s in zip(names, scafs):
            f.write(f">{n} len={len(s)}\n")
            for k in range(0, len(s), width):
                f.write(s[k:k + width] + "\n")


def make_case(root, seed, L=101, frag=(180, 12), jump=(600, 40), n_frag=2500, n_jump=2500, n_contigs=2, end_gap=False, gapless_first=False):
    """Writes scf.fa, result1.sam (frag, --local), result2.sam (jump) and empty dirs; returns the argument lists."""
    rng = np.random.default_rng(np.random.PCG64(seed))
    truths, scafs, gaps = make_scaffolds(rng, n_contigs=n_contigs, end_gap=end_gap, gapless_first=gapless_first)
    names = [f"scf{c}" for c in range(len(scafs))]
    os.makedirs(os.path.join(root, "tmp"), exist_ok=True); os.makedirs(os.path.join(root, "gaps"), exist_ok=True)
    write_fasta(os.path.join(root, "scf.fa"), names, scafs)
    open(os.path.join(root, "result1.sam"), "w").write(make_sam(seed * 7 + 1, truths, scafs, gaps, L, frag[0], frag[1], n_frag, True, names))
    open(os.path.join(root, "result2.sam"), "w").write(make_sam(seed * 7 + 2, truths, scafs, gaps, L, jump[0], jump[1], n_jump, False, names))
    maxd2 = int(jump[0] * 1.15)
    return {"frag": ["scf.fa", str(frag[0]), "1", "result1.sam", "tmp/myout.sam", "scf.fa", "r_1.fastq", "r_2.fastq", "gaps/", "tmp/", "1", "0", "0"],
            "jump": ["scf.fa", str(maxd2), "2", "result2.sam", "tmp/myout.sam", "scf.fa", "r_1.fastq", "r_2.fastq", "gaps/", "tmp/", "1", "0", "0"],
            "n_gaps": None}


if __name__ == "__main__":
    print(make_case(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1))
