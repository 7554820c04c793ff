#!/usr/bin/env python3
"""This container only (needs oracle/_ref): wall time of the host stages either side of the fill on one
larger synthetic input, the reference's own binaries beside the shipped host code, outputs compared byte for byte.
  N1  SAM ingest + binning : oracle/_ref/Preprocess.out  vs  figbird_amd/bin/figtool preprocess   (jump library, samflag 2)
  N2  run-level model      : oracle/_ref/Figbird.out with no gap to fill (parse myout.sam + model, what every worker process
                             of the reference repeats)  vs  libfighost build_model at 1 and at all host threads
  usage: python tools/time_host_stages.py [n_contigs=40] [gaps_per_contig=25] [n_pairs=200000] [out.json] [contig_len=30000] [existing input dir]
  (an existing input dir -- scf.fa + result2.sam of an earlier run -- skips the slow generation; TMPDIR chooses where the
   working copies live: /dev/shm takes this container's slow overlay file system out of the comparison)"""
import filecmp, json, os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from tools import synth_sam, compare_prep
from figbird_amd import api

REF = os.path.join(ROOT, "oracle", "_ref")
TOOL = os.path.join(ROOT, "figbird_amd", "bin", "figtool")


def main():
    n_contigs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    gpc = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    n_pairs = int(sys.argv[3]) if len(sys.argv) > 3 else 200000
    out = sys.argv[4] if len(sys.argv) > 4 else None
    contig_len = int(sys.argv[5]) if len(sys.argv) > 5 else 30000
    L, jump = 101, (600, 40)
    reuse = sys.argv[6] if len(sys.argv) > 6 else None
    base = tempfile.mkdtemp(prefix="fighost_")
    src = os.path.join(base, "src")
    os.makedirs(os.path.join(src, "tmp")); os.makedirs(os.path.join(src, "gaps"))
    t0 = time.time()
    if reuse:
        for fn in ("scf.fa", "result2.sam"):
            os.symlink(os.path.join(os.path.abspath(reuse), fn), os.path.join(src, fn))
        n_rec = int(subprocess.run(["grep", "-vc", "^@", os.path.join(src, "result2.sam")], capture_output=True, text=True).stdout)
        gaps = [None] * 0
        print(f"[host] reusing {reuse}: {n_rec} SAM records", flush=True)
    else:
        rng = np.random.default_rng(np.random.PCG64(4242))
        truths, scafs, gaps = synth_sam.make_scaffolds(rng, n_contigs=n_contigs, contig_len=contig_len, gaps_per_contig=gpc)
        names = [f"scf{c}" for c in range(len(scafs))]
        synth_sam.write_fasta(os.path.join(src, "scf.fa"), names, scafs)
        sam = synth_sam.make_sam(99, truths, scafs, gaps, L, jump[0], jump[1], n_pairs, False, names)
        open(os.path.join(src, "result2.sam"), "w").write(sam)
        n_rec = sum(1 for ln in sam.splitlines() if not ln.startswith("@"))
        print(f"[host] generated {len(gaps)} gaps, {n_rec} SAM records ({len(sam) / 1e6:.0f} MB) in {time.time() - t0:.0f} s", flush=True)
    args = ["scf.fa", str(int(jump[0] * 1.15)), "2", "result2.sam", "tmp/myout.sam", "scf.fa", "r_1.fastq", "r_2.fastq", "gaps/", "tmp/", "1", "0", "0"]
    res, times = {}, {}
    stages = ""
    order = [("figtool", [TOOL, "preprocess"]), ("reference", [os.path.join(REF, "Preprocess.out")]), ("figtool_again", [TOOL, "preprocess"])]
    if n_rec <= 10_000_000:                 # (both sides twice where that is cheap: this container's 8 cores are shared and single runs vary)
        order.append(("reference_again", [os.path.join(REF, "Preprocess.out")]))
    for who, exe in order:
        d = os.path.join(base, who); shutil.copytree(src, d, symlinks=True)
        env = dict(os.environ, FIGSAM_TIMING="1")
        t0 = time.time(); r = subprocess.run(exe + args, cwd=d, capture_output=True, text=True, env=env); times[who] = time.time() - t0
        assert r.returncode == 0, r.stderr[-300:]
        if who == "reference_again":
            shutil.rmtree(d)
            continue
        if who == "figtool_again":
            stages = r.stderr
            assert all(filecmp.cmp(os.path.join(d, f), os.path.join(base, "figtool", f), shallow=False) for f in ("tmp/myout.sam", "tmp/stat.txt"))
            shutil.rmtree(d)
            continue
        res[who] = (compare_prep.outputs(d, "2"), r.stdout)
        print(f"[host] N1 {who}: {times[who]:.2f} s = {n_rec / times[who] / 1e3:.0f} k records/s", flush=True)
    same = res["reference"] == res["figtool"]
    best = min(times["figtool"], times["figtool_again"])
    ref_best = min(times["reference"], times.get("reference_again", times["reference"]))
    line = {"n1_sam_ingest": {"gaps": sum(1 for k in res["figtool"][0] if k.startswith("gaps/gaps_")), "sam_records": n_rec, "reference_s": round(times["reference"], 2),
                              "figtool_s": round(times["figtool"], 2), "figtool_second_run_s": round(times["figtool_again"], 2),
                              "reference_second_run_s": round(times["reference_again"], 2) if "reference_again" in times else None,
                              "speedup": round(times["reference"] / times["figtool"], 1), "speedup_best_of_two": round(ref_best / best, 1),
                              "outputs_identical": bool(same), "files_compared": len(res["reference"][0]), "work_dir": base,
                              "host_threads": os.cpu_count(), "figtool_stage_timers": [ln.strip() for ln in stages.splitlines() if ln.startswith("[figsam]")]}}
    # ---- N2: the model from the myout.sam the ingest just wrote
    d = os.path.join(base, "figtool")
    myout = os.path.join(d, "tmp", "myout.sam")
    n_my = sum(1 for _ in open(myout))
    os.makedirs(os.path.join(d, "gaps"), exist_ok=True)
    open(os.path.join(d, "tmp", "gaploads.txt"), "w").write("\n")
    fig_argv = ["scf.fa", str(int(jump[0] * 1.15)), str(L), "1", "0", "1", "0", "0", "tmp/myout.sam", "tmp/", "gaps/", "30", str(L), "400", "0"]
    print("[host] " + json.dumps(line), flush=True)
    t0 = time.time(); r = subprocess.run([os.path.join(REF, "Figbird.out")] + fig_argv, cwd=d, capture_output=True, text=True); t_ref = time.time() - t0
    ref_ok = r.returncode == 0          # (the reference's worker process dies on some large inputs: recorded, not fatal here)
    tm = {}
    for thr in (1, os.cpu_count() or 8):
        os.environ["FIGFILL_THREADS"] = str(thr)
        t0 = time.time()
        api.model_from_files(os.path.join(d, "scf.fa"), os.path.join(d, "tmp") + "/", myout, partial_flag=0, unmapped_flag=1, script_itr=1,
                             max_distance=int(jump[0] * 1.15), read_length=L, neg_overlap=30, partial_len=L)
        tm[thr] = time.time() - t0
    line["n2_model_build"] = {"myout_records": n_my, "reference_worker_process_s": round(t_ref, 2) if ref_ok else None, "reference_worker_process_rc": r.returncode,
                              "note": "reference: one Figbird.cpp worker process with no gap to fill (scaffold + myout.sam parse + model), repeated by each of its $num_threads processes",
                              **{f"build_model_{k}_threads_s": round(v, 2) for k, v in tm.items()}}
    print("[host] " + json.dumps(line), flush=True)
    if out:
        json.dump(line, open(out, "w"), indent=1)
    shutil.rmtree(base, ignore_errors=True)


if __name__ == "__main__":
    main()
