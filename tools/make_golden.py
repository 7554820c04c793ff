#!/usr/bin/env python3
"""Generate tests/golden/*.tar.gz (this container only; needs oracle/_ref built from /root/reference).

Each fixture = the synthetic post-Preprocess inputs of one seeded case + the outputs the REFERENCE's own
binaries produced on them:
  ref/gapout0.txt, ref/gaptofill0.txt, ref/draw0.txt   <- oracle/_ref/Figbird.out   (Figbird.cpp main, 16 args)
  ref/gapout.txt, ref/filledContigs.fa, ref/Ncount.txt <- oracle/_ref/FillGaps.out  (FillGaps.cpp main, 15 args;
        run in a scratch dir that holds a symlink to /root/reference/Figbird.cpp because it shells out to g++)
Fixtures are data only (inputs + expected outputs); no reference source is copied.
"""
import os, shutil, subprocess, sys, tarfile, tempfile, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from figbird_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
OUT = os.path.join(ROOT, "tests", "golden")

CASES = {
    # name: (kwargs of synth.make_case)
    "unmapped_small": dict(seed=1, mode="unmapped", gap_specs=[(3000, 30), (6000, 600)], coverage=20, n_model_pairs=800),
    "unmapped_mid_err": dict(seed=11, mode="unmapped", gap_specs=[(2500, 45), (5000, 12)], coverage=8, err=0.01, n_model_pairs=800, read_n_rate=0.01),
    "partial_small": dict(seed=2, mode="partial", gap_specs=[(3000, 30), (6000, 120), (9000, 10)], insert_mean=180, insert_sd=10, coverage=30, n_model_pairs=800),
    "partial_brackets": dict(seed=21, mode="partial", gap_specs=[(2000, 3), (3500, 50), (5000, 75), (6500, 101), (8200, 260)], read_len=50,
                             insert_mean=180, insert_sd=10, coverage=12, err=0.005, n_model_pairs=800, contig_len=11000),
    "neg_overlap": dict(seed=31, mode="partial", gap_specs=[(2500, 20), (5000, 10)], insert_mean=180, insert_sd=10, coverage=30, n_model_pairs=800,
                        contig_len=8000, neg_overlap_gaps={1: (10, 14)}),
    "edge_contig_ends": dict(seed=41, mode="unmapped", gap_specs=[(25, 20), (3000, 30), (5960, 25)], contig_len=6000, coverage=10, n_model_pairs=800),
    "edge_no_reads": dict(seed=51, mode="partial", gap_specs=[(2000, 40), (4000, 15)], insert_mean=180, insert_sd=10, coverage=0.0, n_model_pairs=800, contig_len=6000),
    # ---- round 2: reference edge cases of SURVEY §8(c).5 that had no fixture
    "repeat_flanks": dict(seed=61, mode="partial", gap_specs=[(2000, 30), (4000, 40), (6500, 700)], read_len=100, insert_mean=300, insert_sd=15,
                          coverage=14, err=0.002, n_model_pairs=600, contig_len=9500),
    "cap_3001": dict(seed=71, mode="unmapped", gap_specs=[(3000, 25), (6000, 40)], coverage=2400, n_model_pairs=600, max_reads_per_gap=3100),
    "stat2_hint": dict(seed=81, mode="partial", gap_specs=[(2000, 14), (4000, 18), (6000, 30)], insert_mean=180, insert_sd=10, coverage=24, err=0.003,
                       n_model_pairs=800, contig_len=8000),
    "model_indels": dict(seed=91, mode="unmapped", gap_specs=[(3000, 28), (6000, 150)], coverage=10, err=0.01, n_model_pairs=900, model_indel_rate=0.12),
    # $num_threads = 3: FillGaps.cpp:456-649 deals the gaps to three worker processes (gaploads.txt), draw.txt is their
    # draw files one after the other, and each process starts with its own overlap_threshold = 0 (Figbird.cpp:103, :6317)
    "threads3": dict(seed=977, mode="partial", gap_specs=[(1000, 30), (2000, 35), (3000, 24), (4000, 35), (5000, 28), (6000, 40), (7000, 33), (7900, 450)],
                     contig_len=9500, insert_mean=180, insert_sd=10, coverage=18, err=0.003, n_model_pairs=600),
    # ---- round 3: set_inputmean = 1 (Figbird.cpp:6971-6973): pairs on a contig no longer than the library's insert size do not
    # feed the insert-size histogram (:907-914).  The case carries a second, 400-bp scaffold with 260 model pairs of ~300 bp
    # inserts: counted with the flag off (inputMean = 0), dropped with it on, which moves the model's mean / SD / thresholds.
    "inputmean": dict(seed=103, mode="unmapped", gap_specs=[(3000, 30), (6000, 90)], coverage=12, err=0.005, n_model_pairs=700),
    # ---- round 4: the carry of the process-global overlap_threshold (Figbird.cpp:103, :6298-6317; read at :2684, :2760-2766).
    # The FIRST gap of a worker process closes by a negative overlap at its first candidate (gap 0: 10 N, flanks overlapping by
    # 14 bp), so that process's loop leaves at :6306 without getting to :6317; the process's NEXT gap sits 8 bp from the contig
    # end (side_limit < 10, :6303: leaves before :6317 too) and reaches finalize's detect_overlap_gapestimate (:5517) with the
    # global still 0 -- where a carry predicted from contig geometry alone hands it 5.  Its hand-placed partial reads overhang
    # from both sides and overlap by 3 bases (overlap counts of 1..4 are what 0 vs 5 decides, :2684).  `ot_carry`: one worker
    # process {0, 1}; `ot_carry_t3`: $num_threads = 3 deals the four gaps {0, 3}, {1}, {2} (FillGaps.cpp:538-579).
    # (The gap is 52 bp long because such a gap also reads the never-written used_read_arr[0], Figbird.cpp:6265 / :6547: at 52
    # bp the stack residue is 0 in the -O0 build FillGaps compiles and in oracle/_ref's -O2 build alike, which is the value the
    # oracle and the engine take; at most other lengths the reference's two builds disagree with each other.)
    "ot_carry": dict(seed=4, mode="partial", gap_specs=[(2500, 10), (5940, 52)], contig_len=6000, insert_mean=180, insert_sd=10, coverage=30, err=0.003,
                     n_model_pairs=600, neg_overlap_gaps={0: (10, 14)}),
    "ot_carry_t3": dict(seed=5, mode="partial", gap_specs=[(1500, 10), (3000, 30), (4400, 45), (5940, 52)], contig_len=6000, insert_mean=180, insert_sd=10,
                        coverage=30, err=0.003, n_model_pairs=600, neg_overlap_gaps={0: (10, 14)}),
}
N_THREADS = {"threads3": 3, "ot_carry_t3": 3}
SET_INPUTMEAN = {"inputmean": 1}


def _post_edge_no_reads(case):
    for g in case.gaps[:1]:
        g.partial = []


def _flank(case, gi, side, n):
    g = case.gaps[gi]; s = case.scaffolds[0]
    return s[g.start - n:g.start] if side == "L" else s[g.start + g.length:g.start + g.length + n]


def _post_repeat_flanks(case):
    """findRepeat (Figbird.cpp:1799-1911): a partial read that holds a >=21-bp suffix of the 30-bp left flank twice
    AND a >=21-bp prefix of the right flank twice is a two-sided repeat (gap 0: skipped in partial mode, :6183-6186);
    one side only sets one_side_repeat_flag (gap 1: filled with fill=0 semantics; gap 2, > 6 x partial_len long: skipped,
    :6187-6190).  The reads are crafted: tandem repeats of that size cannot come out of a uniform random genome."""
    L = case.read_len
    pad = lambda s: (s + "ACGT" * L)[:L]
    l0, r0 = _flank(case, 0, "L", 24), _flank(case, 0, "R", 24)
    g0 = case.gaps[0]
    two_sided = pad(l0 + l0 + r0 + r0)
    g0.partial.insert(3, synth.PartialRead(two_sided, 23, 1, g0.start - 23, f"24M{L - 24}S", -1, "I" * L))
    for gi in (1, 2):
        g = case.gaps[gi]
        lf = _flank(case, gi, "L", 26)
        one = pad(lf + lf + "TTGACCA")
        g.partial.insert(1, synth.PartialRead(one, 25, 4, g.start - 25, f"26M{L - 26}S", -1, "I" * L))


def _post_cap_3001(case):
    """Preprocess.cpp:1229 lets a gap collect 3001 pairs; Figbird.cpp:7380-7385 then skips it (fillflag=-1)."""
    case.gaps[0].unmapped = case.gaps[0].unmapped[:3001]
    assert len(case.gaps[0].unmapped) == 3001
    case.gaps[1].unmapped = case.gaps[1].unmapped[:36]


def _post_stat2_hint(case):
    """checkMIM's perfect-read hint (Preprocess.cpp:885-925 -> stat2.txt cols 2,3 -> Figbird.cpp:2637): a gap of <= 20 N
    whose hinted length gets the +300 bonus at exactly that candidate length."""
    case.gaps[0].stat2 = (1, 1, 9)            # true length 14: the hint pulls the choice to 9
    case.gaps[1].stat2 = (1, 1, 18)           # hint == true length
    case.gaps[2].stat2 = (1, 1, 25)           # G0 = 30 > 20: the hint must be ignored


def _post_threads3(case):
    """Gaps 1-3 get only a few hand-placed partial reads whose left- and right-hanging ends overlap by 2-4 bases: the overlaps
    detect_overlap_gapestimate (Figbird.cpp:2512-2700) judges against overlap_threshold, which is 0 for the first gap of a
    worker process and 5 for the later ones."""
    s = case.scaffolds[0]; L = case.read_len
    for gi, (into_left, clip_right) in {1: ((20, 16), (18, 12)), 3: ((19, 15), (18, 14)), 2: ((14, 10), (12, 9))}.items():
        g = case.gaps[gi]
        reads = []
        for k in into_left:
            al = L - k; pos1 = g.start - al + 1
            reads.append(synth.PartialRead(s[g.start - al:g.start] + g.truth[:k], g.start - pos1, 1, pos1, f"{al}M{k}S", -1, "I" * L))
        for c in clip_right:
            al = L - c
            reads.append(synth.PartialRead(g.truth[len(g.truth) - c:] + s[g.start + g.length:g.start + g.length + al], c, 2, g.start + g.length + 1, f"{c}S{al}M", -1, "I" * L))
        g.partial = reads


def _post_ot_carry(case):
    """The last gap (8 bp before the contig end): right-anchored reads (8 aligned bases, all the flank there is) covering its last
    L - 8 bases and left-anchored reads hanging far enough into the gap to overlap them by 3 and by 0 bases."""
    s = case.scaffolds[0]; L = case.read_len
    g = case.gaps[-1]
    assert len(s) - (g.start + g.length) == 8
    reads = []
    c_right = L - 8
    for k in (g.length - c_right + 3, g.length - c_right, g.length - c_right + 3):
        al = L - k; pos1 = g.start - al + 1
        reads.append(synth.PartialRead(s[g.start - al:g.start] + g.truth[:k], g.start - pos1, 1, pos1, f"{al}M{k}S", -1, "I" * L))
    for c in (c_right, c_right):
        al = L - c
        reads.append(synth.PartialRead(g.truth[len(g.truth) - c:] + s[g.start + g.length:g.start + g.length + al], c, 2, g.start + g.length + 1, f"{c}S{al}M", -1, "I" * L))
    g.partial = reads


def _post_inputmean(case):
    rng = synth.np.random.default_rng(synth.np.random.PCG64(1031))
    short = synth._rand_seq(rng, 400)
    case.scaffolds.append(short); case.truth.append(short)
    L = case.read_len
    for k in range(260):
        isz = int(round(rng.normal(300, 8)))
        p0 = int(rng.integers(0, 400 - isz))
        r1 = short[p0:p0 + L]; r2 = short[p0 + isz - L:p0 + isz]
        q = f"s1_{k}"
        case.myout.append("\t".join([q, "99", "1", str(p0 + 1), f"{L}M", str(isz), r1, "I" * L, f"MD:Z:{L}", "IH:i:1"]))
        case.myout.append("\t".join([q, "147", "1", str(p0 + isz - L + 1), f"{L}M", str(-isz), r2, "I" * L, f"MD:Z:{L}", "IH:i:1"]))
    case.n_pairs = len(case.myout) // 2


POST = {"ot_carry": _post_ot_carry, "ot_carry_t3": _post_ot_carry, "inputmean": _post_inputmean, "threads3": _post_threads3, "edge_no_reads": _post_edge_no_reads, "repeat_flanks": _post_repeat_flanks, "cap_3001": _post_cap_3001, "stat2_hint": _post_stat2_hint}


def make(name):
    """The Case of a golden fixture, exactly as it was generated (tests rebuild in-memory batches from this)."""
    kw = dict(CASES[name])
    case = synth.make_case(name, kw.pop("seed"), kw.pop("mode"), kw.pop("gap_specs"), **kw)
    if name in POST:
        POST[name](case)
    return case


def build(name, kw=None, keep=None):
    case = make(name)
    base = tempfile.mkdtemp(prefix="figgold_")
    root = os.path.join(base, name)
    p = synth.write_case(case, root)
    synth.write_gaploads(p, list(range(len(case.gaps))))
    refdir = os.path.join(root, "ref")
    os.makedirs(refdir)
    sim = SET_INPUTMEAN.get(name, 0)
    r = subprocess.run([os.path.join(REF, "Figbird.out")] + synth.figbird_argv(case, p, set_inputmean=sim), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for fn in ("gapout0.txt", "gaptofill0.txt", "draw0.txt"):
        shutil.move(p["tmp"] + fn, os.path.join(refdir, fn))
    os.remove(p["tmp"] + "gaploads.txt")
    # FillGaps.out shells out to `g++ Figbird.cpp`: give it a scratch cwd with a symlink to the reference source
    cwd = os.path.join(base, "cwd"); os.makedirs(cwd)
    os.symlink("/root/reference/Figbird.cpp", os.path.join(cwd, "Figbird.cpp"))
    nthr = N_THREADS.get(name, 1)
    r = subprocess.run([os.path.join(REF, "FillGaps.out")] + synth.fillgaps_argv(case, p, n_threads=nthr, set_inputmean=sim), capture_output=True, text=True, cwd=cwd)
    assert r.returncode == 0, r.stderr
    for fn in ("gapout.txt", "filledContigs.fa", "Ncount.txt", "draw.txt"):
        shutil.move(p["tmp"] + fn, os.path.join(refdir, fn))
    if nthr > 1:
        shutil.move(p["tmp"] + "gaploads.txt", os.path.join(refdir, "gaploads.txt"))      # FillGaps.cpp:313-334, as the reference left it
    else:
        os.remove(p["tmp"] + "gaploads.txt")
    meta = {"name": name, "figbird_argv": ["scf.fa"] + synth.figbird_argv(case, p)[1:8] + ["tmp/myout.sam", "tmp/", "gaps/"] + synth.figbird_argv(case, p, set_inputmean=sim)[11:],
            "fillgaps_argv": ["scf.fa"] + synth.fillgaps_argv(case, p, n_threads=nthr)[1:7] + ["tmp/myout.sam", "tmp/", "gaps/"] + synth.fillgaps_argv(case, p, set_inputmean=sim)[10:],
            "mode": case.mode, "n_gaps": len(case.gaps), "truth": [g.truth for g in case.gaps]}
    with open(os.path.join(root, "meta.json"), "w") as f:
        json.dump(meta, f)
    os.makedirs(OUT, exist_ok=True)
    tgz = os.path.join(OUT, name + ".tar.gz")
    with tarfile.open(tgz, "w:gz") as t:
        t.add(root, arcname=name)
    shutil.rmtree(base)
    return tgz


if __name__ == "__main__":
    for name in CASES:
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        print(build(name), flush=True)
