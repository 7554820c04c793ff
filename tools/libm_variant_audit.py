#!/usr/bin/env python3
"""Dev tool (this container only, CPU): is "glibc-exact" one target?  And does the reference care?

glibc 2.35 selects its `exp` / `log` / `pow` (and with `log`, `log10`) through IFUNCs: an FMA-contracted build on hosts with FMA + AVX2,
a plain SSE2 build elsewhere.  `GLIBC_TUNABLES=glibc.cpu.hwcaps=-FMA,-AVX2_Usable` forces the second on a host that has the
first, so both can be run here.

Part 1 evaluates the four calls the path makes (`Figbird.cpp:3169,3179,3591,3601`: log10 / log of a likelihood product, pow(10, t) /
exp of a log-domain weight) on N seeded arguments of the path's ranges under both variants and counts the results that differ
and by how many ulps: the reference's OWN weights move from host to host by this much.

Part 2 runs the reference's worker program itself (`oracle/_ref/Figbird.out`, compiled from /root/reference where it lies by
oracle/Makefile) on the seeded fuzz cases of tools/fuzz_ref.py (general + mid-bracket) under both variants and compares
gapout0.txt / gaptofill0.txt / draw0.txt byte for byte: a flip = a case whose bytes differ between the two glibc variants.

usage: python3 tools/libm_variant_audit.py <N args> <first seed> <count> [mid_first mid_count] [workers=8]
Last line: a JSON record."""
import json, os, shutil, subprocess, sys, tempfile, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from figbird_amd import synth
from tools.fuzz_ref import mk, mk_mid
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "Figbird.out")
FILES = ("gapout0.txt", "gaptofill0.txt", "draw0.txt")
NOFMA = "glibc.cpu.hwcaps=-FMA,-AVX2_Usable,-FMA4"

ARGS_C = r"""
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
static uint64_t s = 88172645463325252ull;
static double u01(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (s >> 11) * (1.0 / 9007199254740992.0); }
int main(int argc, char **argv) {
    long n = atol(argv[1]);
    FILE *f = fopen(argv[2], "wb");
    static double o[4 * 4096];
    for (long i = 0; i < n; i += 4096) {
        for (int k = 0; k < 4096; k++) {
            double p = exp(-700.0 * u01());            /* a product of up to ~150 per-base probabilities: (0, 1] over 300 decades */
            double t = -300.0 * u01();                 /* a log10-domain weight relative to the read's maximum */
            o[4 * k + 0] = log10(p);
            o[4 * k + 1] = log(p);
            o[4 * k + 2] = pow(10.0, t);
            o[4 * k + 3] = exp(t * 2.302585092994046);
        }
        fwrite(o, sizeof(double), 4 * 4096, f);
    }
    fclose(f);
    return 0;
}
"""


def env_of(nofma):
    env = dict(os.environ)
    env.pop("GLIBC_TUNABLES", None)
    if nofma: env["GLIBC_TUNABLES"] = NOFMA
    return env


def part1(n):
    d = tempfile.mkdtemp(prefix="figlv_")
    try:
        open(os.path.join(d, "lv.c"), "w").write(ARGS_C)
        subprocess.run(["gcc", "-O2", "-fno-builtin", "-o", os.path.join(d, "lv"), os.path.join(d, "lv.c"), "-lm"], check=True)
        for tag, nofma in (("a", False), ("b", True)):
            subprocess.run([os.path.join(d, "lv"), str(n), os.path.join(d, tag + ".bin")], check=True, env=env_of(nofma))
        a = np.fromfile(os.path.join(d, "a.bin"), dtype=np.int64).reshape(-1, 4)
        b = np.fromfile(os.path.join(d, "b.bin"), dtype=np.int64).reshape(-1, 4)
        out = {"args": int(a.shape[0])}
        for i, nm in enumerate(("log10", "log", "pow10", "exp")):
            df = np.abs(a[:, i] - b[:, i])
            out[nm] = {"differ": int((df != 0).sum()), "max_ulp": int(df.max())}
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def run_ref(case, root, nofma):
    p = synth.write_case(case, root)
    synth.write_gaploads(p, list(range(len(case.gaps))))
    r = subprocess.run([REF] + synth.figbird_argv(case, p), capture_output=True, text=True, env=env_of(nofma))
    if r.returncode != 0: raise RuntimeError(f"reference rc {r.returncode}: {r.stderr[-200:]}")
    return tuple(open(p["tmp"] + fn, "rb").read() if os.path.exists(p["tmp"] + fn) else None for fn in FILES)


def one_case(job):
    kind, gen, seed = job
    case = gen(seed)
    base = tempfile.mkdtemp(prefix="figlv_")
    try:
        a = run_ref(case, os.path.join(base, "fma"), False)
        b = run_ref(case, os.path.join(base, "sse2"), True)
        return kind, seed, case.mode, len(case.gaps), [fn for fn, x, y in zip(FILES, a, b) if x != y]
    finally:
        shutil.rmtree(base, ignore_errors=True)


def main():
    a = sys.argv[1:]
    n, first, count = int(a[0]), int(a[1]), int(a[2])
    mid_first, mid_count = (int(a[3]), int(a[4])) if len(a) > 4 else (0, 0)
    workers = int(a[5]) if len(a) > 5 else 8
    t0 = time.time()
    p1 = part1(n)
    print("part 1 (results that differ between glibc's FMA and SSE2 variants):", p1, flush=True)
    jobs = [("general", mk, s) for s in range(first, first + count)] + [("mid-bracket", mk_mid, s) for s in range(mid_first, mid_first + mid_count)]
    flips, gaps, modes = [], 0, {}
    with ThreadPoolExecutor(max_workers=workers) as ex:
        for kind, seed, mode, ng, diff in ex.map(one_case, jobs):
            gaps += ng; modes[mode] = modes.get(mode, 0) + 1
            if diff:
                flips.append([kind, seed, diff]); print(f"FLIP {kind} seed {seed} ({mode}): {diff}", flush=True)
    rec = {"tool": "libm_variant_audit", "glibc": os.confstr("CS_GNU_LIBC_VERSION"), "tunable": NOFMA, "libm_results": p1,
           "reference_cases": len(jobs), "gaps": gaps, "modes": modes, "cases_whose_bytes_differ": len(flips), "flips": flips,
           "seconds": round(time.time() - t0, 1)}
    print(f"{len(jobs)} cases ({gaps} gaps) through oracle/_ref/Figbird.out under both variants: {len(flips)} differ")
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
