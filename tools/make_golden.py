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
}


def build(name, kw, keep=None):
    kw = dict(kw)
    case = synth.make_case(name, kw.pop("seed"), kw.pop("mode"), kw.pop("gap_specs"), **kw)
    if name == "edge_no_reads":
        for g in case.gaps[:1]:
            g.partial = []
    base = tempfile.mkdtemp(prefix="figgold_")
    root = os.path.join(base, name)
    p = synth.write_case(case, root)
    synth.write_gaploads(p, list(range(len(case.gaps))))
    refdir = os.path.join(root, "ref")
    os.makedirs(refdir)
    r = subprocess.run([os.path.join(REF, "Figbird.out")] + synth.figbird_argv(case, p), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for fn in ("gapout0.txt", "gaptofill0.txt", "draw0.txt"):
        shutil.move(p["tmp"] + fn, os.path.join(refdir, fn))
    os.remove(p["tmp"] + "gaploads.txt")
    # FillGaps.out shells out to `g++ Figbird.cpp`: give it a scratch cwd with a symlink to the reference source
    cwd = os.path.join(base, "cwd"); os.makedirs(cwd)
    os.symlink("/root/reference/Figbird.cpp", os.path.join(cwd, "Figbird.cpp"))
    r = subprocess.run([os.path.join(REF, "FillGaps.out")] + synth.fillgaps_argv(case, p, n_threads=1), capture_output=True, text=True, cwd=cwd)
    assert r.returncode == 0, r.stderr
    for fn in ("gapout.txt", "filledContigs.fa", "Ncount.txt", "draw.txt"):
        shutil.move(p["tmp"] + fn, os.path.join(refdir, fn))
    os.remove(p["tmp"] + "gaploads.txt")
    meta = {"name": name, "figbird_argv": ["scf.fa"] + synth.figbird_argv(case, p)[1:8] + ["tmp/myout.sam", "tmp/", "gaps/"] + synth.figbird_argv(case, p)[11:],
            "fillgaps_argv": ["scf.fa"] + synth.fillgaps_argv(case, p)[1:7] + ["tmp/myout.sam", "tmp/", "gaps/"] + synth.fillgaps_argv(case, p)[10:],
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
    for name, kw in CASES.items():
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        print(build(name, kw), flush=True)
