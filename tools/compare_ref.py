#!/usr/bin/env python3
"""Dev/fixture tool (this container only): run oracle/_ref and oracle/figbird_oracle on the
same synthetic case and diff gapout / gaptofill / filledContigs.fa / Ncount.txt."""
import os, shutil, subprocess, sys, time, tempfile, filecmp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from figbird_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
ORA = os.path.join(ROOT, "oracle", "figbird_oracle")


def run_ref_figbird(case, root):
    p = synth.write_case(case, root)
    synth.write_gaploads(p, list(range(len(case.gaps))))
    t = time.time()
    r = subprocess.run([os.path.join(REF, "Figbird.out")] + synth.figbird_argv(case, p), capture_output=True, text=True)
    return p, time.time() - t, r


def run_oracle_figbird(case, root, trace=None, level=1):
    p = synth.write_case(case, root)
    synth.write_gaploads(p, list(range(len(case.gaps))))
    env = dict(os.environ)
    if trace:
        env["FIG_ORACLE_TRACE"] = trace
        env["FIG_ORACLE_TRACE_LEVEL"] = str(level)
    t = time.time()
    r = subprocess.run([ORA, "figbird"] + synth.figbird_argv(case, p), capture_output=True, text=True, env=env)
    return p, time.time() - t, r


def compare(case, base, verbose=True):
    shutil.rmtree(base, ignore_errors=True)
    pr, tr, rr = run_ref_figbird(case, os.path.join(base, "ref"))
    po, to, ro = run_oracle_figbird(case, os.path.join(base, "ora"))
    ok = True
    for fn in ("gapout0.txt", "gaptofill0.txt", "draw0.txt"):
        a = open(pr["tmp"] + fn).read() if os.path.exists(pr["tmp"] + fn) else None
        b = open(po["tmp"] + fn).read() if os.path.exists(po["tmp"] + fn) else None
        same = a == b
        ok &= same
        if verbose and not same:
            print(f"  DIFF {fn}")
            if a and b:
                for i, (x, y) in enumerate(zip(a.splitlines(), b.splitlines())):
                    if x != y:
                        print("   ref:", x[:300]); print("   ora:", y[:300]); break
    if verbose:
        print(f"{case.name}: {'OK' if ok else 'MISMATCH'} ref {tr:.2f}s oracle {to:.2f}s rc {rr.returncode}/{ro.returncode} {ro.stderr[-200:]}")
    return ok


if __name__ == "__main__":
    base = tempfile.mkdtemp(prefix="figcmp_")
    cases = [
        synth.make_case("u_small", 1, "unmapped", [(3000, 30), (6000, 600)], coverage=20),
        synth.make_case("p_small", 2, "partial", [(3000, 30), (6000, 120), (9000, 10)], insert_mean=180, insert_sd=10, coverage=30),
    ]
    allok = True
    for c in cases:
        allok &= compare(c, os.path.join(base, c.name))
    shutil.rmtree(base, ignore_errors=True)
    sys.exit(0 if allok else 1)
