#!/usr/bin/env python3
"""tests/golden/pipeline_e2e.tar.gz (this container only): one pass through every stage of RunFigbird.sh's schedule on
synthetic data with the aligner stubbed by tools/synth_sam.py -- BASELINE config 1's shape (one scaffold set, a ~180-bp
frag library; plus one jump-library iteration) at toy size:

  iteration 1 (frag, partial mode, RunFigbird.sh:253-360):  FlankTrim -> reference.py -> [bowtie2 --local: stub] ->
      Preprocess samflag 1 -> FillGaps (partial_flag=1)
  iteration 2 (jump, unmapped mode, :263-360) on the same trimmed scaffold:  [bowtie2: stub] Preprocess samflag 1, then
      Preprocess samflag 2 -> FillGaps (unmapped=1)
  CombineGaps over gapout_1.txt, gapout_2.txt (:777)

Every stage is the REFERENCE's own program (oracle/_ref/*.out, /root/reference/reference.py); the fixture stores the
inputs a stage cannot regenerate (scaffold, SAM files) and every stage's outputs.  tests/test_pipeline.py replays the
schedule with figtool / figfill and compares each file."""
import json, os, shutil, subprocess, sys, tarfile, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import synth_sam
from figbird_amd import synth

REF = os.path.join(ROOT, "oracle", "_ref")
OUT = os.path.join(ROOT, "tests", "golden")
L, TRIM = 101, 10


def n_runs(s):
    out, i = [], 0
    while i < len(s):
        if s[i] in "Nn":
            j = i
            while j < len(s) and s[j] in "Nn":
                j += 1
            out.append((i, j - i)); i = j
        else:
            i += 1
    return out


def build(root, seed=5):
    rng = np.random.default_rng(np.random.PCG64(seed))
    os.makedirs(root)
    t = synth._rand_seq(rng, 5200)
    # truth intervals cut out: (truth start, true length, N-run length)
    cuts = [(900, 9, 9), (2300, 16, 14), (3700, 24, 24)]
    parts, cur, gaps0 = [], 0, []
    for ts, tl, nl in cuts:
        parts.append(t[cur:ts]); parts.append("N" * nl); cur = ts + tl
    parts.append(t[cur:])
    scaf = "".join(parts)
    synth_sam.write_fasta(os.path.join(root, "draft.fa"), ["scf0"], [scaf])
    exp = os.path.join(root, "expected"); os.makedirs(exp)
    # ---- FlankTrim + reference.py (iteration 1 only)
    subprocess.run([os.path.join(REF, "FlankTrim.out"), os.path.join(root, "draft.fa"), str(TRIM), str(L), os.path.join(exp, "trimmed_temp.fa")], check=True)
    subprocess.run([sys.executable, "/root/reference/reference.py", os.path.join(exp, "trimmed_temp.fa"), os.path.join(exp, "trimmed.fa"), "60"], check=True)
    trimmed = "".join(l.strip() for l in open(os.path.join(exp, "trimmed_temp.fa")) if not l.startswith(">"))
    runs = n_runs(trimmed)
    assert len(runs) == len(cuts)
    gaps = []
    off = 0                                                   # scaffold - truth offset left of the gap
    for (ts, tl, nl), (ss, sl) in zip(cuts, runs):
        grow = (sl - nl) // 2                                 # bases N-masked either side
        gaps.append((0, ss, sl, tl + 2 * grow, ts - grow))
        off += nl - tl
    # ---- the aligner stub: reads from the truth, records against the TRIMMED scaffold
    frag = synth_sam.make_sam(seed * 7 + 1, [t], [trimmed], gaps, L, 180, 12, 1500, True, ["scf0"])
    jump = synth_sam.make_sam(seed * 7 + 2, [t], [trimmed], gaps, L, 600, 40, 500, False, ["scf0"])
    open(os.path.join(root, "result1.sam"), "w").write(frag)
    open(os.path.join(root, "result2.sam"), "w").write(jump)
    meta = {"L": L, "trim": TRIM, "frag_isz": 180, "jump_isz": 600, "jump_maxd": int(600 * 1.15), "truth": t, "cuts": cuts}
    json.dump(meta, open(os.path.join(root, "meta.json"), "w"))

    def stage_dir(name):
        d = os.path.join(exp, name); os.makedirs(os.path.join(d, "tmp")); os.makedirs(os.path.join(d, "gaps")); return d

    def prep(d, maxd, samflag, sam):
        r = subprocess.run([os.path.join(REF, "Preprocess.out"), os.path.join(exp, "trimmed.fa"), str(maxd), str(samflag), os.path.join(root, sam), "tmp/myout.sam",
                            os.path.join(exp, "trimmed.fa"), "r_1.fastq", "r_2.fastq", "gaps/", "tmp/", "1", "0", "0"], cwd=d, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr

    def fill(d, maxd, count, pflag, uflag, isz):
        cwd = tempfile.mkdtemp(); os.symlink("/root/reference/Figbird.cpp", os.path.join(cwd, "Figbird.cpp"))
        a = [os.path.join(exp, "trimmed.fa"), str(maxd), str(L), str(count), str(pflag), str(uflag), "1", os.path.join(d, "tmp/myout.sam"), os.path.join(d, "tmp") + "/",
             os.path.join(d, "gaps") + "/", "30", str(L), str(TRIM), "0", str(isz)]
        r = subprocess.run([os.path.join(REF, "FillGaps.out")] + a, cwd=cwd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        for fn in os.listdir(os.path.join(d, "tmp")):
            if fn.startswith("gaploads"):
                os.remove(os.path.join(d, "tmp", fn))

    d1 = stage_dir("iter1"); prep(d1, 180, 1, "result1.sam"); fill(d1, 180, 1, 1, 0, 180)
    d2 = stage_dir("iter2"); prep(d2, 180, 1, "result1.sam")
    shutil.copytree(d2, os.path.join(exp, "iter2_after_frag"))
    prep(d2, meta["jump_maxd"], 2, "result2.sam"); fill(d2, meta["jump_maxd"], 2, 0, 1, 600)
    cg = os.path.join(exp, "combine"); os.makedirs(cg)
    shutil.copy(os.path.join(d1, "tmp", "gapout.txt"), os.path.join(cg, "gapout_1.txt")); shutil.copy(os.path.join(d2, "tmp", "gapout.txt"), os.path.join(cg, "gapout_2.txt"))
    subprocess.run([os.path.join(REF, "CombineGaps.out"), "2", cg + "/"], check=True)


if __name__ == "__main__":
    base = tempfile.mkdtemp(prefix="figpipe_")
    root = os.path.join(base, "pipeline_e2e")
    import time; t0 = time.time()
    build(root)
    print("reference schedule took", round(time.time() - t0, 1), "s")
    tgz = os.path.join(OUT, "pipeline_e2e.tar.gz")
    with tarfile.open(tgz, "w:gz") as t:
        t.add(root, arcname="pipeline_e2e")
    print(tgz, os.path.getsize(tgz))
    shutil.rmtree(base)
