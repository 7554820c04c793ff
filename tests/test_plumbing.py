"""The stages either side of the fill, restated in the C++ host (figtool: SURVEY.md §8f N1, N3, N4), against golden
outputs of the reference's own programs (tools/make_plumb_golden.py) and -- where oracle/_ref exists -- against those
programs run live on fresh synthetic inputs."""
import json
import os
import shutil
import subprocess

import pytest

import util
from tools.compare_prep import mask_gap_file

HAVE_REF = os.path.exists(os.path.join(util.ROOT, "oracle", "_ref", "Preprocess.out")) and os.path.exists("/root/reference/reference.py")


def _tool(args, cwd):
    return subprocess.run([util.FIGTOOL] + args, cwd=cwd, capture_output=True, text=True)


def _same(a, b):
    assert open(a, "rb").read() == open(b, "rb").read(), (a, b)


def test_rewrap_59_60_61(tmp_path):
    """reference.py:5-28: 60-column chunks each with a newline, the remainder without; a blank line after a sequence whose
    length is a multiple of 60; no newline at the end of the file."""
    root = util.extract_golden("plumbing", str(tmp_path))
    assert _tool(["rewrap", "rewrap_in.fa", "out.fa", "60"], root).returncode == 0
    _same(os.path.join(root, "out.fa"), os.path.join(root, "expected", "rewrap_out.fa"))
    out = open(os.path.join(root, "out.fa")).read().split(">")
    assert out[1].split("\n")[1:] == [out[1].split("\n")[1], ""] and len(out[1].split("\n")[1]) == 59      # 59: one short line, then the header's newline
    assert out[2].split("\n")[2] == ""                                                                       # 60: chunk + blank line
    assert not open(os.path.join(root, "out.fa")).read().endswith("\n")


@pytest.mark.parametrize("name", ["genome_oneline.fa", "genome_wrapped.fa", "genome_nonl.fa"])
def test_flanktrim_and_reduce_match_reference_outputs(name, tmp_path):
    root = util.extract_golden("plumbing", str(tmp_path))
    for trim in (10, 0, 3):
        assert _tool(["flanktrim", name, str(trim), "101", f"ft_{trim}.fa"], root).returncode == 0
        _same(os.path.join(root, f"ft_{trim}.fa"), os.path.join(root, "expected", f"flanktrim_{trim}_{name}"))
    os.makedirs(os.path.join(root, "t"))
    assert _tool(["reduce-scf", name, "t/"], root).returncode == 0
    _same(os.path.join(root, "t", "newgenome.fa"), os.path.join(root, "expected", f"reduce_{name}"))


def test_trim_then_rewrap_chain(tmp_path):
    """RunFigbird.sh:254-256: FlankTrim's one-line-per-contig output goes through reference.py."""
    root = util.extract_golden("plumbing", str(tmp_path))
    assert _tool(["flanktrim", "genome_oneline.fa", "10", "101", "t.fa"], root).returncode == 0
    assert _tool(["rewrap", "t.fa", "tw.fa", "60"], root).returncode == 0
    _same(os.path.join(root, "tw.fa"), os.path.join(root, "expected", "rewrap_trimmed.fa"))


@pytest.mark.parametrize("n", [1, 2, 3])
def test_combine_gaps_matches_reference_outputs(n, tmp_path):
    root = util.extract_golden("plumbing", str(tmp_path))
    d = os.path.join(root, f"c{n}"); os.makedirs(d)
    for k in range(1, n + 1):
        shutil.copy(os.path.join(root, "combine", f"gapout_{k}.txt"), d)
    assert _tool(["combine", str(n), f"c{n}/"], root).returncode == 0
    _same(os.path.join(d, "combined_gapstring.txt"), os.path.join(root, "expected", f"combined_gapstring_{n}.txt"))
    _same(os.path.join(d, "Individual_gaps.txt"), os.path.join(root, "expected", f"Individual_gaps_{n}.txt"))


def test_missing_inputs_fail_like_the_reference(tmp_path):
    assert _tool(["flanktrim", "nope.fa", "10", "101", "o.fa"], str(tmp_path)).returncode == 1
    assert _tool(["reduce-scf", "nope.fa", "./"], str(tmp_path)).returncode == 1
    assert not os.path.exists(os.path.join(str(tmp_path), "newgenome.fa"))
    assert _tool(["combine", "1", "./"], str(tmp_path)).returncode == 1


def _prep_outputs(root):
    d = {}
    for fn in ("tmp/gapInfo.txt", "tmp/stat.txt", "tmp/stat2.txt", "tmp/myout.sam"):
        d[fn] = open(os.path.join(root, fn)).read()
    for fn in sorted(os.listdir(os.path.join(root, "gaps"))):
        t = open(os.path.join(root, "gaps", fn)).read()
        d["gaps/" + fn] = mask_gap_file(t) if fn.startswith("gaps_") else t
    return d


@pytest.mark.parametrize("lib", ["frag", "jump"])
def test_preprocess_boundary_golden(lib, tmp_path):
    """SAM ingest + binning (Preprocess.cpp): gapInfo/stat/stat2, myout.sam and the per-gap read files byte-identical to the
    reference's; in gaps_<g>.sam the two fields the reference fills from uninitialised memory are masked (see compare_prep)."""
    root = util.extract_golden("preprocess_boundary", str(tmp_path))
    args = json.load(open(os.path.join(root, "args.json")))[lib]
    os.makedirs(os.path.join(root, "tmp")); os.makedirs(os.path.join(root, "gaps"))
    r = _tool(["preprocess"] + args, root)
    assert r.returncode == 0, r.stderr
    got = _prep_outputs(root)
    exp = _prep_outputs(os.path.join(root, "expected_" + lib))
    assert sorted(got) == sorted(exp)
    for k in exp:
        assert got[k] == exp[k], k
    assert r.stdout == open(os.path.join(root, "expected_" + lib, "stdout.txt")).read()
    n_reads = sum(v.count("\n") for k, v in got.items() if k.startswith("gaps/"))
    assert n_reads > 50                                     # the fixture really bins reads


@pytest.mark.parametrize("seed,threads", [(201, 1), (202, 3), (203, 7)])
def test_preprocess_fast_and_legacy_forms_agree(seed, threads, tmp_path):
    """The ingest's fast form (mapped SAM parsed once in parallel chunks, pair logic on the parsed records, myout.sam formatted
    straight into a mapping of the output file, hash-indexed duplicate test) against the line-by-line legacy form
    (FIGSAM_LEGACY=1) on fresh synthetic SAMs of both libraries: every output file byte-identical, whatever the chunking."""
    import shutil
    from tools import synth_sam
    src = str(tmp_path / "src")
    args = synth_sam.make_case(src, seed, n_contigs=1 + seed % 3, end_gap=(seed % 2 == 0), n_frag=700, n_jump=1500)
    for lib in ("frag", "jump"):
        outs = {}
        for form, env in (("fast", {"FIGFILL_THREADS": str(threads)}), ("legacy", {"FIGSAM_LEGACY": "1"})):
            d = str(tmp_path / f"{lib}_{form}"); shutil.copytree(src, d)
            a = list(args[lib]); a[-1] = "0"
            r = util.run([util.FIGTOOL, "preprocess"] + a, d, env)
            assert r.returncode == 0, r.stderr
            outs[form] = (_prep_outputs(d), r.stdout)
        assert sorted(outs["fast"][0]) == sorted(outs["legacy"][0])
        for k in outs["legacy"][0]:
            assert outs["fast"][0][k] == outs["legacy"][0][k], (lib, k)
        assert outs["fast"][1] == outs["legacy"][1]


@pytest.mark.skipif(not HAVE_REF, reason="needs the reference binaries (build container only)")
@pytest.mark.parametrize("seed", [101, 102, 103])
def test_preprocess_live_against_reference(seed, tmp_path):
    from tools.compare_prep import run_one
    assert run_one(seed, str(tmp_path), verbose=False, end_gap=(seed % 3 == 0), n_contigs=1 + seed % 3, n_frag=900, n_jump=900)


@pytest.mark.skipif(not os.path.exists("/root/reference/RunFigbird.sh"), reason="needs the reference's driver script (this container only)")
def test_driver_patch_applies_to_the_reference_script(tmp_path):
    """integration/RunFigbird.patch -- the edits INTEGRATION.md §1/§4b describe (every `g++ X.cpp && ./a.out` and
    `python reference.py` of the gap-fill pipeline re-pointed at figfill / figtool) -- applies cleanly to the reference's
    RunFigbird.sh, leaves a script bash parses, and touches nothing but those 14 command lines (+ the FIGBIRD_AMD guard)."""
    import shutil, subprocess
    dst = tmp_path / "RunFigbird.sh"
    shutil.copy("/root/reference/RunFigbird.sh", dst)
    patch = os.path.join(util.ROOT, "integration", "RunFigbird.patch")
    r = subprocess.run(["patch", "-s", str(dst), patch], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert subprocess.run(["bash", "-n", str(dst)]).returncode == 0
    new = dst.read_text().split("\n"); old = open("/root/reference/RunFigbird.sh").read().split("\n")
    assert len(new) == len(old) + 2
    changed = [i for i in range(len(old)) if old[i] != new[i + 2 if i >= 2 else i]]
    assert [i + 1 for i in changed] == [254, 256, 266, 285, 320, 338, 352, 433, 435, 451, 472, 480, 777, 809]
    txt = "\n".join(new)
    assert txt.count('"$FIGBIRD_AMD/bin/figfill"') == 2 and txt.count('"$FIGBIRD_AMD/bin/figtool"') == 12
    assert "g++ -std=c++11 -pthread FillGaps.cpp" not in txt and "g++ Preprocess.cpp" not in txt
