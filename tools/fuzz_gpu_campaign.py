#!/usr/bin/env python3
"""Dev tool (GPU box): a longer run of the two GPU fuzz generators the test suite samples (tests/test_gpu_parity.py):
fresh seeded inputs through bin/figfill on the device against the oracle on the host, every output file compared byte for byte.

usage: python3 tools/fuzz_gpu_campaign.py <first seed> <count> [mid_first mid_count]
Prints one line per seed and a summary; exit status 1 on any mismatch."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.fuzz_ref import mk, mk_mid
from tools.compare_emu import run_one
from tests import util

def main():
    a = [int(x) for x in sys.argv[1:]]
    first, count = a[0], a[1]
    mid_first, mid_count = (a[2], a[3]) if len(a) >= 4 else (0, 0)
    bad = []
    t0 = time.time()
    for kind, gen, seeds in (("general", mk, range(first, first + count)), ("mid-bracket", mk_mid, range(mid_first, mid_first + mid_count))):
        for seed in seeds:
            with tempfile.TemporaryDirectory() as d:
                t1 = time.time()
                ok = bool(run_one(gen(seed), d, exe=util.FIGFILL, verbose=False))
            print("%s seed %d: %s (%.1f s)" % (kind, seed, "identical" if ok else "MISMATCH", time.time() - t1), flush=True)
            if not ok: bad.append((kind, seed))
    print("# %d general + %d mid-bracket seeds, %d mismatches %s, %.0f s" % (count, mid_count, len(bad), bad, time.time() - t0))
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
