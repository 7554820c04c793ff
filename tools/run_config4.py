#!/usr/bin/env python3
"""GPU box: a BASELINE config-4-shaped run on ONE MI355X -- a scaffold set with 10^4 gaps (GAGE gap mix), 2x100-bp
jump-library reads, one unmapped-mode fill pass through the C ABI -- as evidence that the packer, the per-gap slabs and the
candidate-parallel scheduler hold at that scale.  Reports gaps/s, filled bases, the share of called bases that equal the
synthetic truth, and a byte comparison of a small stratified sample with the oracle (test infrastructure, here only as
the checker).
  usage: python tools/run_config4.py [n_gaps=10000] [read_len=100] [out.json]"""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from figbird_amd import api, synth


def main():
    if os.environ.get("FIG_PROBE_TORCH"):          # A/B: torch (and the HIP runtime it brings) initialised first, as in bench.py
        import torch
        torch.cuda.init(); torch.zeros(1, device="cuda")
    n_gaps = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    read_len = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    out = sys.argv[3] if len(sys.argv) > 3 else None
    insert = float(os.environ.get("FIG_CFG4_INSERT", "2500"))      # 3500 = the bench's jump library
    spec = synth.BenchSpec(mode="unmapped", read_len=read_len, insert_mean=insert, insert_sd=insert / 10, reads_per_gap_mean=1000.0,
                           frag_len=min(101, read_len))      # partial reads longer than the run's maxReadLength are outside the envelope (FIG_EUNSUP)
    mc = synth.bench_model_case(7, spec)
    work = tempfile.mkdtemp(prefix="figcfg4_")
    mp = synth.write_case(mc, os.path.join(work, "model"))
    model = api.model_from_files(mp["scf"], mp["tmp"], mp["myout"], partial_flag=0, unmapped_flag=1, script_itr=1,
                                 max_distance=spec.max_distance, read_length=spec.read_len, neg_overlap=30, partial_len=mc.partial_len)
    t0 = time.time(); batch, truth = synth.make_bench_batch(int(os.environ.get("FIG_CFG4_SEED", "4004")), n_gaps, spec); t_gen = time.time() - t0
    n_reads = int(batch.u_read_off[-1])
    print(f"[config4] {n_gaps} gaps, {n_reads} reads generated in {t_gen:.1f} s", flush=True)
    eng = api.Engine(0)
    eng.set_model(model)
    sub = int(os.environ.get("FIG_CFG4_SUBBATCH", "0"))
    if sub > 0:
        # the same gaps as consecutive fills of `sub` gaps each (dealt round-robin in file order, so every fill sees the same mix)
        tot_ms = tot_fl = 0.0; filled = 0; t0 = time.time()
        for k in range(0, (n_gaps + sub - 1) // sub):
            ids = list(range(k, n_gaps, (n_gaps + sub - 1) // sub))
            sb = synth.subset_batch(batch, ids)
            r = eng.fill(sb); st = eng.stats()
            tot_ms += st["kernel_ms"]; tot_fl += st["alg_flops"]; filled += int(r.filled_bases)
            print(f"[config4] sub-batch {k}: {len(ids)} gaps, kernels {st['kernel_ms'] / 1e3:.1f} s", flush=True)
        line = {"workload": f"{n_gaps} gaps as {(n_gaps + sub - 1) // sub} consecutive fills of ~{sub}", "kernel_ms": round(tot_ms, 1), "wall_s": round(time.time() - t0, 1),
                "gaps_per_s": round(n_gaps / (tot_ms / 1e3), 2), "alg_tflops": round(tot_fl / 1e12 / (tot_ms / 1e3), 3),
                "frac_of_fp64_nofma_peak": round(tot_fl / 1e12 / (tot_ms / 1e3) / 39.3, 4), "filled_bases": filled}
        print("[config4] " + json.dumps(line), flush=True)
        if out:
            json.dump(line, open(out, "w"), indent=1)
        eng.close()
        return
    t0 = time.time(); eng.upload(batch); t_up = time.time() - t0
    print(f"[config4] packed + uploaded in {t_up:.1f} s", flush=True)
    # the fill is one blocking ABI call of several minutes: a heartbeat keeps the log moving meanwhile
    import threading
    stop = threading.Event()
    def beat():
        t = time.time()
        while not stop.wait(45.0):
            print(f"[config4] filling ... {time.time() - t:.0f} s", flush=True)
    th = threading.Thread(target=beat, daemon=True); th.start()
    t0 = time.time(); res = eng.fill_resident(); t_fill = time.time() - t0
    stop.set(); th.join()
    st = eng.stats()
    called = mism = 0
    for s, t in zip(res.strings, truth):
        if len(s) != len(t):
            continue
        a = np.frombuffer(s.encode(), dtype=np.uint8)
        m = a != ord("N")
        called += int(m.sum()); mism += int((a[m] != t[m]).sum())
    line = {"workload": f"config-4 shape: {n_gaps} gaps (GAGE mix), 2x{read_len}-bp jump reads (insert {int(insert)}+-{int(insert / 10)}), {n_reads} reads, one unmapped-mode fill on 1 MI355X",
            "n_gaps": n_gaps, "n_reads": n_reads, "gen_s": round(t_gen, 2), "pack_upload_s": round(t_up, 2), "fill_wall_s": round(t_fill, 2),
            "kernel_ms": round(st["kernel_ms"], 1), "gaps_per_s": round(n_gaps / max(st["kernel_ms"] / 1e3, 1e-9), 2),
            "filled_bases": int(res.filled_bases), "filled_bases_per_s": round(res.filled_bases / max(st["kernel_ms"] / 1e3, 1e-9), 1),
            "alg_tflops": round(st["alg_flops"] / 1e12 / max(st["kernel_ms"] / 1e3, 1e-9), 3), "frac_of_fp64_nofma_peak": round(st["alg_flops"] / 1e12 / max(st["kernel_ms"] / 1e3, 1e-9) / 39.3, 4),
            "place_calls": int(st["place_calls"]), "called_bases": called, "called_bases_equal_truth": called - mism,
            "packed_bytes": int(st.get("packed_bytes", 0))}
    print("[config4] " + json.dumps(line), flush=True)
    # ---- byte comparison of a stratified sample with the oracle (cheapest gap of each bracket + the most expensive <=30-bp one)
    oracle = os.path.join(ROOT, "oracle", "figbird_oracle")
    if os.path.exists(oracle):
        G = np.asarray(batch.gap_len); nr = np.diff(batch.u_read_off)
        sample = []
        for lo, hi in [(5, 31), (31, 134), (401, 800), (800, 1300), (1300, 2001)]:
            ids = [g for g in range(n_gaps) if lo <= G[g] < hi]
            if ids:
                sample.append(min(ids, key=lambda g: (int(nr[g]) * (int(G[g]) if G[g] <= 400 else 1), g)))
        paths = synth.write_batch_subset(batch, sample, mc, os.path.join(work, "cpu"), spec)
        args = [paths["scf"], str(spec.max_distance), str(spec.read_len), "1", "0", "1", "1", paths["myout"], paths["tmp"], paths["gaps"],
                "30", str(mc.partial_len), "10", "0", str(int(spec.insert_mean))]
        t0 = time.time(); r = subprocess.run([oracle, "fillgaps"] + args, cwd=work, capture_output=True, text=True, timeout=900)
        ok = r.returncode == 0
        if ok:
            lines = open(paths["tmp"] + "gapout.txt").read().splitlines()
            for k, g in enumerate(paths["gap_order"]):
                if g in sample:
                    f = lines[k].split("\t")
                    ok = ok and int(f[4]) == int(res.filled_len[g]) and (f[5] if len(f) > 5 else "") == res.strings[g]
        line["oracle_sample"] = {"gaps": [int(g) for g in sample], "gap_lengths": [int(G[g]) for g in sample], "identical": bool(ok), "oracle_s": round(time.time() - t0, 1)}
        print("[config4] oracle sample: " + json.dumps(line["oracle_sample"]), flush=True)
    if out:
        with open(out, "w") as f:
            json.dump(line, f, indent=1)
    eng.close()


if __name__ == "__main__":
    main()
