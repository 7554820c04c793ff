#!/usr/bin/env python3
"""Dev tool: fuzz figfill (emu by default, FIGFILL_EXE to override) against the oracle."""
import os, sys, tempfile, shutil
from concurrent.futures import ProcessPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.fuzz_ref import mk
from tools.compare_emu import run_one, EMU

def one(seed):
    base = tempfile.mkdtemp(prefix=f"figfe{seed}_")
    try:
        c = mk(seed)
        ok = run_one(c, base, exe=os.environ.get("FIGFILL_EXE", EMU), verbose=False)
        return seed, ok, c.mode, [g.length for g in c.gaps]
    except Exception as e:
        return seed, False, "EXC " + repr(e), []
    finally:
        shutil.rmtree(base, ignore_errors=True)

if __name__ == "__main__":
    a, b = int(sys.argv[1]), int(sys.argv[2])
    bad = []
    with ProcessPoolExecutor(max_workers=int(os.environ.get("FIG_WORKERS", "8"))) as ex:
        for seed, ok, mode, gl in ex.map(one, range(a, b)):
            print(seed, "OK" if ok else "MISMATCH", mode, gl, flush=True)
            if not ok: bad.append(seed)
    print("bad seeds:", bad)
