#!/usr/bin/env python3
"""Dev tool: fuzz the oracle restatement against oracle/_ref over many seeded cases."""
import os, sys, tempfile, shutil, random
from concurrent.futures import ProcessPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from figbird_amd import synth
from tools.compare_ref import compare

def mk(seed):
    rnd = random.Random(seed)
    mode = rnd.choice(["unmapped", "partial", "partial"])
    L = rnd.choice([36, 50, 76, 101])
    err = rnd.choice([0.0, 0.005, 0.02])
    if mode == "partial":
        gaps = []
        pos = 1500
        for _ in range(rnd.randint(2, 5)):
            g = rnd.choice([1, 3, 8, 15, 25, L - 5, L, L + 10, 2 * L, 2 * L + 30, 5 * L])
            gaps.append((pos, g)); pos += g + rnd.randint(800, 1500)
        neg = {}
        if rnd.random() < 0.4:
            gi = rnd.randrange(len(gaps)); neg[gi] = (rnd.choice([5, 10, 20]), rnd.choice([6, 10, 14, 20]))
        return synth.make_case(f"fz{seed}", seed, "partial", gaps, contig_len=pos + 1500, read_len=L,
                               insert_mean=rnd.choice([180, 220]), insert_sd=10, coverage=rnd.choice([3, 10, 30]),
                               err=err, n_model_pairs=1500, neg_overlap_gaps=neg, neg_overlap=rnd.choice([30, 0]),
                               script_itr=rnd.choice([1, 2]))
    else:
        gaps = []
        pos = 1500
        for _ in range(rnd.randint(1, 3)):
            g = rnd.choice([5, 12, 20, 29, 30, 45, 420, 600, 900])
            gaps.append((pos, g)); pos += g + rnd.randint(900, 1500)
        return synth.make_case(f"fz{seed}", seed, "unmapped", gaps, contig_len=pos + 1500, read_len=L,
                               insert_mean=rnd.choice([400, 600]), insert_sd=rnd.choice([20, 40]),
                               coverage=rnd.choice([4, 12, 25]), err=err, n_model_pairs=1500,
                               partial_reads_in_unmapped=rnd.random() < 0.7, read_n_rate=rnd.choice([0, 0, 0.01]),
                               script_itr=rnd.choice([1, 2]))

def mk_mid(seed):
    """Unmapped cases of the 134-400 bracket (candidate range [0.5 G0, 2.5 G0], Figbird.cpp:6894-6901; large_gap_flag off, up
    to ~400 candidate lengths per gap), which mk() leaves out for its CPU cost: one gap, few reads."""
    rnd = random.Random(seed)
    L = rnd.choice([36, 50])
    g = rnd.choice([134, 150, 180, 230, 300, 400])
    return synth.make_case(f"fzm{seed}", seed, "unmapped", [(1500, g)], contig_len=1500 + g + 1500, read_len=L,
                           insert_mean=rnd.choice([400, 600]), insert_sd=rnd.choice([20, 40]), coverage=rnd.choice([3, 5]),
                           err=rnd.choice([0.0, 0.01]), n_model_pairs=1200, partial_reads_in_unmapped=rnd.random() < 0.7,
                           read_n_rate=rnd.choice([0, 0.01]), script_itr=1)


def one(seed):
    base = tempfile.mkdtemp(prefix=f"figfz{seed}_")
    try:
        c = mk(seed)
        ok = compare(c, base, verbose=False)
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
