#!/usr/bin/env python3
"""Generate tests/golden/plumbing.tar.gz and tests/golden/preprocess_boundary.tar.gz (this container only): synthetic
inputs + the outputs of the REFERENCE's own stages on them -- oracle/_ref/{FlankTrim,Reduce_SCF,CombineGaps,Preprocess}.out
(compiled from /root/reference where the sources lie) and /root/reference/reference.py run as the script it is.
Fixtures are data (inputs and expected outputs) only."""
import os, random, shutil, subprocess, sys, tarfile, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import synth_sam

REF = os.path.join(ROOT, "oracle", "_ref")
OUT = os.path.join(ROOT, "tests", "golden")


def rnd(rng, n, alpha="ACGT"):
    return "".join(rng.choice(alpha) for _ in range(n))


def plumbing_inputs(root, seed=1):
    rng = random.Random(seed)
    os.makedirs(root, exist_ok=True)
    # ---- rewrap: 59 / 60 / 61 columns, an empty record, two lines' worth, and a last line without newline
    with open(os.path.join(root, "rewrap_in.fa"), "w") as f:
        for n, l in (("a", 59), ("b", 60), ("c", 61), ("d", 0), ("e", 120), ("f", 7)):
            f.write(f">{n} some text\n" + rnd(rng, l, "ACGTN") + "\n")
        f.write(">g\n" + rnd(rng, 75))
    # ---- flanktrim / reduce: contigs on one line each (as FillGaps writes them) and 60-column wrapped, gaps of many kinds
    def contig(n, gaps):
        s = list(rnd(rng, n))
        for st, ln, ch in gaps:
            s[st:st + ln] = ch * ln
        return "".join(s)
    cs = [("scfA extra words", contig(3000, [(400, 5, "N"), (900, 1, "N"), (1400, 40, "N"), (1900, 150, "N"), (2500, 30, "n"), (2990, 10, "N")])),
          ("scfB", contig(700, [])),
          ("scfC", contig(1500, [(20, 8, "N"), (300, 12, "N"), (330, 9, "N"), (800, 60, "N")])),
          ("scfD", ""),
          ("scfE", contig(2100, [(1000, 25, "N")]).replace("A", "a", 3))]
    with open(os.path.join(root, "genome_oneline.fa"), "w") as f:
        for n, s in cs:
            f.write(f">{n}\n{s}\n")
    with open(os.path.join(root, "genome_wrapped.fa"), "w") as f:
        for n, s in cs:
            f.write(f">{n}\n")
            for k in range(0, len(s), 60):
                f.write(s[k:k + 60] + "\n")
    # a line longer than the 1023-character fgets buffer is what the one-line form gives; also one without trailing newline
    with open(os.path.join(root, "genome_nonl.fa"), "w") as f:
        f.write(">x\n" + contig(1200, [(500, 20, "N")]) + "\n>y\n" + contig(400, [(100, 6, "N")]))
    # ---- combine: three iterations of gapout files; gaps close, stay partly open, close to length 0, or stay untouched
    g1 = [(0, 0, 100, 30, rnd(rng, 30)), (1, 0, 400, 200, rnd(rng, 60) + "N" * 80 + rnd(rng, 60)), (2, 0, 900, 10, ""),
          (3, 1, 50, 500, "N" * 500), (4, 1, 700, 45, rnd(rng, 20) + "N" * 25), (5, 1, 1200, 12, "N" * 5 + rnd(rng, 7))]
    g2 = [(0, 0, 460, 80, rnd(rng, 25) + "N" * 30 + rnd(rng, 25)), (1, 1, 50, 500, rnd(rng, 100) + "N" * 300 + rnd(rng, 100)), (2, 1, 720, 25, rnd(rng, 25)), (3, 1, 1200, 5, rnd(rng, 5))]
    g3 = [(0, 0, 485, 30, rnd(rng, 30)), (1, 1, 150, 300, rnd(rng, 140) + "N" * 20 + rnd(rng, 140))]
    d = os.path.join(root, "combine"); os.makedirs(d, exist_ok=True)
    for k, g in enumerate((g1, g2, g3), 1):
        with open(os.path.join(d, f"gapout_{k}.txt"), "w") as f:
            for (no, c, st, g0, s) in g:
                f.write(f"{no}\t{c}\t{st}\t{g0}\t{len(s)}\t{s}\n")


def run_ref_plumbing(root):
    exp = os.path.join(root, "expected"); os.makedirs(exp, exist_ok=True)
    subprocess.run([sys.executable, "/root/reference/reference.py", os.path.join(root, "rewrap_in.fa"), os.path.join(exp, "rewrap_out.fa"), "60"], check=True)
    for name in ("genome_oneline.fa", "genome_wrapped.fa", "genome_nonl.fa"):
        for trim in (10, 0, 3):
            subprocess.run([os.path.join(REF, "FlankTrim.out"), os.path.join(root, name), str(trim), "101", os.path.join(exp, f"flanktrim_{trim}_{name}")], check=True)
        t = tempfile.mkdtemp()
        subprocess.run([os.path.join(REF, "Reduce_SCF.out"), os.path.join(root, name), t + "/"], check=True)
        shutil.move(os.path.join(t, "newgenome.fa"), os.path.join(exp, f"reduce_{name}"))
    # the trimmed genome goes through reference.py next (RunFigbird.sh:254-256)
    subprocess.run([sys.executable, "/root/reference/reference.py", os.path.join(exp, "flanktrim_10_genome_oneline.fa"), os.path.join(exp, "rewrap_trimmed.fa"), "60"], check=True)
    for n in (1, 2, 3):
        t = tempfile.mkdtemp()
        for k in range(1, n + 1):
            shutil.copy(os.path.join(root, "combine", f"gapout_{k}.txt"), t)
        subprocess.run([os.path.join(REF, "CombineGaps.out"), str(n), t + "/"], check=True)
        shutil.move(os.path.join(t, "combined_gapstring.txt"), os.path.join(exp, f"combined_gapstring_{n}.txt"))
        shutil.move(os.path.join(t, "Individual_gaps.txt"), os.path.join(exp, f"Individual_gaps_{n}.txt"))


def preprocess_case(root, seed=11):
    args = synth_sam.make_case(root, seed, n_frag=1300, n_jump=1300, n_contigs=2, end_gap=False)
    # a third, gap-free contig + the reduced genome (Reduce_SCF) for the genome_reduction=1 variant
    for lib in ("frag", "jump"):
        d = tempfile.mkdtemp()
        for fn in ("scf.fa", "result1.sam", "result2.sam"):
            shutil.copy(os.path.join(root, fn), d)
        os.makedirs(os.path.join(d, "tmp")); os.makedirs(os.path.join(d, "gaps"))
        r = subprocess.run([os.path.join(REF, "Preprocess.out")] + args[lib], cwd=d, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        dst = os.path.join(root, "expected_" + lib)
        os.makedirs(dst)
        shutil.move(os.path.join(d, "tmp"), os.path.join(dst, "tmp")); shutil.move(os.path.join(d, "gaps"), os.path.join(dst, "gaps"))
        open(os.path.join(dst, "stdout.txt"), "w").write(r.stdout)
    import json
    json.dump({"frag": args["frag"], "jump": args["jump"]}, open(os.path.join(root, "args.json"), "w"))
    shutil.rmtree(os.path.join(root, "tmp")); shutil.rmtree(os.path.join(root, "gaps"))


def pack(root, name):
    os.makedirs(OUT, exist_ok=True)
    tgz = os.path.join(OUT, name + ".tar.gz")
    with tarfile.open(tgz, "w:gz") as t:
        t.add(root, arcname=name)
    print(tgz, os.path.getsize(tgz))


if __name__ == "__main__":
    base = tempfile.mkdtemp(prefix="figplumb_")
    p = os.path.join(base, "plumbing"); plumbing_inputs(p); run_ref_plumbing(p); pack(p, "plumbing")
    q = os.path.join(base, "preprocess_boundary"); preprocess_case(q); pack(q, "preprocess_boundary")
    shutil.rmtree(base)
