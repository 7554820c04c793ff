#!/usr/bin/env python3
"""Dev tool (CPU): how much of the path's output hangs on the last bit of a libm result?

The device evaluates log / pow(10, .) / log10 / exp with its own routines, which differ from glibc's by at most 1 ulp
(tools/ubench/*_check.c); everything else in the chain is IEEE-identical.  The parity tests have never
seen a decision flip because of it -- this tool turns that into a measured bound.  For every seeded fuzz case (the generators of
tools/fuzz_ref.py: partial and unmapped mode, the mid-bracket regime) the oracle runs once as it is and then J times with
FIG_ORACLE_ULP_JITTER=<j>[:k] (oracle/figbird_oracle.cpp: EVERY result of those four calls moved by a random whole number of
ulps in [-k, k] -- far more often than the device differs from glibc, which is in ~0.4 % of the logarithms and never by more than
one ulp), and the four output files are compared byte for byte.  A flip = a (case, jitter) run whose bytes differ.

usage: python3 tools/libm_jitter_audit.py <first seed> <count> <jitter runs per case> [k=1] [mid_first mid_count] [workers=8] [permille=1000]
(permille < 1000: only that share of the libm calls is moved -- 10 is about twice the rate at which the device differs from glibc)
Prints one line per case with a flip, a summary line, and (last line) a JSON record."""
import json, os, shutil, subprocess, sys, tempfile, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from figbird_amd import synth
from tools.fuzz_ref import mk, mk_mid
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORA = os.path.join(ROOT, "oracle", "figbird_oracle")
FILES = ("gapout.txt", "filledContigs.fa", "Ncount.txt", "draw.txt")


def outputs(case, d, jitter):
    p = synth.write_case(case, d)
    env = dict(os.environ)
    env.pop("FIG_ORACLE_ULP_JITTER", None)
    if jitter is not None: env["FIG_ORACLE_ULP_JITTER"] = jitter
    r = subprocess.run([ORA, "fillgaps"] + synth.fillgaps_argv(case, p), capture_output=True, text=True, env=env)
    if r.returncode != 0: raise RuntimeError(f"oracle rc {r.returncode}: {r.stderr[-200:]}")
    return tuple(open(p["tmp"] + fn, "rb").read() if os.path.exists(p["tmp"] + fn) else None for fn in FILES)


def one_case(args):
    kind, gen, seed, J, k, pm = args
    case = gen(seed)
    base = tempfile.mkdtemp(prefix="figjit_")
    try:
        ref = outputs(case, os.path.join(base, "ref"), None)
        flips = []
        for j in range(1, J + 1):
            got = outputs(case, os.path.join(base, f"j{j}"), f"{seed * 1000 + j}:{k}:{pm}")
            if got != ref: flips.append((j, [fn for fn, a, b in zip(FILES, ref, got) if a != b]))
            shutil.rmtree(os.path.join(base, f"j{j}"), ignore_errors=True)
        n_gaps = len(case.gaps) if hasattr(case, "gaps") else 0
        return kind, seed, getattr(case, "mode", "?"), n_gaps, flips
    finally:
        shutil.rmtree(base, ignore_errors=True)


def main():
    a = sys.argv[1:]
    first, count, J = int(a[0]), int(a[1]), int(a[2])
    k = int(a[3]) if len(a) > 3 else 1
    mid_first, mid_count = (int(a[4]), int(a[5])) if len(a) > 5 else (0, 0)
    workers = int(a[6]) if len(a) > 6 else 8
    pm = int(a[7]) if len(a) > 7 else 1000
    jobs = [("general", mk, s, J, k, pm) for s in range(first, first + count)] + [("mid-bracket", mk_mid, s, J, k, pm) for s in range(mid_first, mid_first + mid_count)]
    t0 = time.time()
    n_runs = n_flip_runs = n_flip_cases = 0
    per_mode = {}
    with ThreadPoolExecutor(max_workers=workers) as ex:
        for kind, seed, mode, n_gaps, flips in ex.map(one_case, jobs):
            n_runs += J; n_flip_runs += len(flips); n_flip_cases += bool(flips)
            m = per_mode.setdefault(mode, {"cases": 0, "runs": 0, "flip_runs": 0}); m["cases"] += 1; m["runs"] += J; m["flip_runs"] += len(flips)
            if flips: print(f"{kind} seed {seed} ({mode}): {len(flips)} of {J} jittered runs differ: {flips[:3]}", flush=True)
    rec = {"cases": len(jobs), "jitter_runs": n_runs, "k_ulps": k, "share_of_calls_permille": pm, "flip_runs": n_flip_runs, "cases_with_a_flip": n_flip_cases, "per_mode": per_mode,
           "seeds": {"general": [first, count], "mid_bracket": [mid_first, mid_count]}, "seconds": round(time.time() - t0)}
    print(f"# {len(jobs)} cases x {J} jittered runs ({pm / 10:g} % of the libm results moved by up to {k} ulp): {n_flip_runs} runs with different bytes, in {n_flip_cases} cases, {rec['seconds']} s")
    print(json.dumps(rec))
    return 0


if __name__ == "__main__":
    sys.exit(main())
