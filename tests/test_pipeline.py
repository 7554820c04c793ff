"""One pass through every stage of RunFigbird.sh's schedule (BASELINE config 1's shape at toy size; aligner stubbed by
tools/synth_sam.py) with the host tools of this repo in place of the reference's programs, every stage's files compared
with what the reference's own programs wrote (tests/golden/pipeline_e2e.tar.gz, tools/make_pipeline_golden.py):
FlankTrim -> reference.py -> Preprocess (frag) -> FillGaps (partial) ; Preprocess (frag, jump) -> FillGaps (unmapped) ;
CombineGaps.  The fill runs on the one-lane emulation here and on the MI355X in the -m gpu variant."""
import json
import os
import shutil
import subprocess

import pytest

import util
from tools.compare_prep import mask_gap_file


def _run(cmd, cwd, env=None):
    e = dict(os.environ); e.update(env or {})
    r = subprocess.run(cmd, cwd=cwd, capture_output=True, text=True, env=e)
    assert r.returncode == 0, (cmd, r.stderr[-400:])
    return r


def _cmp_dir(got, exp, names, mask=False):
    for fn in names:
        a, b = open(os.path.join(got, fn)).read(), open(os.path.join(exp, fn)).read()
        if mask and os.path.basename(fn).startswith("gaps_"):
            a, b = mask_gap_file(a), mask_gap_file(b)
        assert a == b, fn


def _schedule(figfill_exe, tmp_path):
    root = util.extract_golden("pipeline_e2e", str(tmp_path))
    exp = os.path.join(root, "expected")
    m = json.load(open(os.path.join(root, "meta.json")))
    L, trim, maxd2 = str(m["L"]), str(m["trim"]), str(m["jump_maxd"])
    tool = util.FIGTOOL
    # ---- iteration 1 preamble: FlankTrim + rewrap (RunFigbird.sh:253-256)
    _run([tool, "flanktrim", "draft.fa", trim, L, "trimmed_temp.fa"], root)
    _run([tool, "rewrap", "trimmed_temp.fa", "trimmed.fa", "60"], root)
    _cmp_dir(root, exp, ["trimmed_temp.fa", "trimmed.fa"])

    def stage(name):
        d = os.path.join(root, name); os.makedirs(os.path.join(d, "tmp")); os.makedirs(os.path.join(d, "gaps")); return d

    def prep(d, maxd, samflag, sam):
        _run([tool, "preprocess", "../trimmed.fa", maxd, samflag, "../" + sam, "tmp/myout.sam", "../trimmed.fa", "r_1.fastq", "r_2.fastq", "gaps/", "tmp/", "1", "0", "0"], d)

    def fill(d, maxd, count, pflag, uflag, isz, env=None):
        _run([figfill_exe, "../trimmed.fa", maxd, L, count, pflag, uflag, "1", "tmp/myout.sam", "tmp/", "gaps/", "30", L, trim, "0", isz], d, env)

    prep_files = ["tmp/gapInfo.txt", "tmp/stat.txt", "tmp/stat2.txt", "tmp/myout.sam"]
    fill_files = ["tmp/gapout.txt", "tmp/draw.txt", "tmp/filledContigs.fa", "tmp/Ncount.txt"]
    # ---- iteration 1: frag library, partial mode
    d1 = stage("iter1")
    prep(d1, "180", "1", "result1.sam")
    _cmp_dir(d1, os.path.join(exp, "iter1"), prep_files + ["gaps/" + f for f in sorted(os.listdir(os.path.join(exp, "iter1", "gaps")))])
    fill(d1, "180", "1", "1", "0", "180")
    _cmp_dir(d1, os.path.join(exp, "iter1"), fill_files)
    # ---- iteration 2: frag SAM for the soft-clipped reads, jump SAM for the mate-anchored ones, unmapped mode
    d2 = stage("iter2")
    prep(d2, "180", "1", "result1.sam")
    _cmp_dir(d2, os.path.join(exp, "iter2_after_frag"), prep_files)
    prep(d2, maxd2, "2", "result2.sam")
    _cmp_dir(d2, os.path.join(exp, "iter2"), prep_files + ["gaps/" + f for f in sorted(os.listdir(os.path.join(exp, "iter2", "gaps")))], mask=True)
    fill(d2, maxd2, "2", "0", "1", "600")
    _cmp_dir(d2, os.path.join(exp, "iter2"), fill_files)
    # ---- the same iteration with the SAM ingested inside figfill and the reads handed over in memory (no per-gap files)
    d3 = stage("iter2_mem")
    fill(d3, maxd2, "2", "0", "1", "600", env={"FIGFILL_SAM": f"180:../result1.sam;{maxd2}:../result2.sam"})
    _cmp_dir(d3, os.path.join(exp, "iter2"), fill_files + prep_files)
    assert os.listdir(os.path.join(d3, "gaps")) == []
    # ---- CombineGaps over the two iterations' gapout files (RunFigbird.sh:354-360, 777)
    cg = os.path.join(root, "cg"); os.makedirs(cg)
    shutil.copy(os.path.join(d1, "tmp", "gapout.txt"), os.path.join(cg, "gapout_1.txt"))
    shutil.copy(os.path.join(d2, "tmp", "gapout.txt"), os.path.join(cg, "gapout_2.txt"))
    _run([tool, "combine", "2", "cg/"], root)
    _cmp_dir(cg, os.path.join(exp, "combine"), ["combined_gapstring.txt", "Individual_gaps.txt"])
    # the fixture is not trivial: gaps got filled and both read kinds were binned
    assert any(len(l.split("\t")) > 5 and l.split("\t")[5].strip("N\n") for l in open(os.path.join(d2, "tmp", "gapout.txt")))
    assert sum(os.path.getsize(os.path.join(d2, "gaps", f)) for f in os.listdir(os.path.join(d2, "gaps"))) > 1000


def test_schedule_on_the_emulation(tmp_path):
    _schedule(util.EMU, tmp_path)


@pytest.mark.gpu
def test_schedule_on_the_device(tmp_path):
    _schedule(util.FIGFILL, tmp_path)
