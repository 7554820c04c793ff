// figfill -- drop-in for the reference's gap-fill stage.  RunFigbird.sh:352/:480 runs
//   g++ -std=c++11 -pthread FillGaps.cpp && ./a.out <15 args>
// figfill takes the same 15 positional arguments (FillGaps.cpp:419-433), reads the same
// files Preprocess.cpp left behind and writes the same outputs (Temp/filledContigs.fa,
// gapout.txt, draw.txt, Ncount.txt) -- but instead of fanning out `g++ Figbird.cpp`
// processes it builds the model once, packs every gap into one batch and hands it to the
// MI355X engine through the C ABI of include/figbird_hip.h.  Exit codes follow the
// reference: message on stderr + exit(1).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <chrono>
#include <thread>

#include "fig_host.h"
#include "fig_sam.h"

using namespace fighost;

static int fail(const std::string &m) { fprintf(stderr, "%s\n", m.c_str()); return 1; }

// ---- N GPUs of one node from this process (FIGFILL_DEVICES=0,1,...; the reference starts its own workers too,
// FillGaps.cpp:668-679): the gaps are dealt longest-processing-time-first on estimated cost (fig_host.h; the role of
// FillGaps.cpp:456-649), one host thread per GPU creates its own fig_ctx (contexts are independent, include/figbird_hip.h)
// and fills its shard through fig_fill_gaps with no exchange on the data path; the shard results meet in host memory in
// global gap / read order, so the writers below see exactly what a single-GPU run hands them.
struct ShardRun { std::vector<int64_t> ids; Batch sub; Results R; fig_stats st; int rc = 0; std::string err; fig_ctx *ctx = nullptr; std::vector<uint8_t> reach, preset; };

static void shard_fail(ShardRun *S, int rc, const std::string &what) {
    S->rc = rc; S->err = what + ": " + fig_strerror(rc);
    if (S->ctx) { fig_ctx_destroy(S->ctx); S->ctx = nullptr; }
}

// phase 1: context, model, upload, and which of the shard's gaps get to Figbird.cpp:6317 (fig_batch_probe_reach)
static void shard_open(int device, const fig_model *fm, const Scaffold *sc, ShardRun *S) {
    memset(&S->st, 0, sizeof(S->st));
    fig_gap_batch fb; S->sub.view(fb, *sc);
    const int64_t ng = fb.n_gaps;
    S->reach.assign((size_t)std::max<int64_t>(ng, 1), 0);
    if (ng == 0) return;
    int rc = fig_ctx_create(device, &S->ctx);
    if (rc) { S->ctx = nullptr; return shard_fail(S, rc, std::string("fig_ctx_create(") + std::to_string(device) + ")"); }
    if ((rc = fig_ctx_set_model(S->ctx, fm))) return shard_fail(S, rc, "fig_ctx_set_model");
    if ((rc = fig_batch_upload(S->ctx, &fb))) return shard_fail(S, rc, std::string("fig_batch_upload on device ") + std::to_string(device));
    if ((rc = fig_batch_probe_reach(S->ctx, S->reach.data()))) return shard_fail(S, rc, "fig_batch_probe_reach");
}

// phase 2: the carry worked out over ALL shards (S->preset, shard order), then the fill
static void shard_fill(int device, const fig_model *fm, const Scaffold *sc, ShardRun *S) {
    fig_gap_batch fb; S->sub.view(fb, *sc);
    const int64_t ng = fb.n_gaps;
    if (ng == 0 || !S->ctx) return;
    int rc = fig_batch_set_ot_preset(S->ctx, S->preset.data());
    if (rc) return shard_fail(S, rc, "fig_batch_set_ot_preset");
    Results &R = S->R;
    const int64_t cap = fig_results_capacity(fm, &fb);
    R.filled_len.assign(ng, 0); R.gaptofill.assign(ng, 0); R.str_off.assign(ng + 1, 0); R.str.assign((size_t)std::max<int64_t>(cap, 1), 'N');
    const int64_t nr = (int64_t)S->sub.u_anchor_pos.size() + (int64_t)S->sub.p_pos.size();
    R.draw_pos.assign((size_t)std::max<int64_t>(nr, 1), INT32_MIN); R.draw_isz.assign((size_t)std::max<int64_t>(nr, 1), 0); R.draw_len.assign((size_t)std::max<int64_t>(ng * 2, 1), -1);
    fig_gap_results fr; memset(&fr, 0, sizeof(fr));
    fr.filled_len = R.filled_len.data(); fr.gaptofill = R.gaptofill.data(); fr.str_off = R.str_off.data();
    fr.str = &R.str[0]; fr.str_capacity = (int64_t)R.str.size();
    fr.draw_pos = R.draw_pos.data(); fr.draw_isz = R.draw_isz.data(); fr.draw_len = R.draw_len.data();
    rc = fig_fill_resident(S->ctx, &fr);
    fig_get_stats(S->ctx, &S->st);
    if (rc) return shard_fail(S, rc, std::string("fig_fill_resident on device ") + std::to_string(device));
    fig_ctx_destroy(S->ctx); S->ctx = nullptr;
}

// Fill B over `devices`; on success R holds the whole gap set in global order.  Returns 0 or 1 (message in err).
static int fill_multi(const std::vector<int> &devices, const RunArgs &a, const Scaffold &sc, Batch &B, const fig_model &fm, Results &R,
                      fig_stats &st, std::string &err) {
    const int world = (int)devices.size();
    const int64_t ng = (int64_t)B.gap_contig.size();
    std::vector<std::vector<int64_t>> shards = partition_lpt(estimate_cost(B, a, fm.max_read_length), world);
    std::vector<ShardRun> runs((size_t)world);
    for (int r = 0; r < world; r++) { runs[r].ids = shards[r]; make_shard(B, shards[r], runs[r].sub); }
    const bool serial = getenv("FIGFILL_SERIAL") != nullptr;   // test knob: one shard after the other (the CPU emulation library keeps global state)
    auto phase = [&](void (*fn)(int, const fig_model *, const Scaffold *, ShardRun *)) {
        if (serial) { for (int r = 0; r < world; r++) fn(devices[r], &fm, &sc, &runs[r]); return; }
        std::vector<std::thread> th;
        for (int r = 0; r < world; r++) th.emplace_back(fn, devices[r], &fm, &sc, &runs[r]);
        for (auto &t : th) t.join();
    };
    auto failed = [&]() { for (int r = 0; r < world; r++) if (runs[r].rc) { err = "figfill: " + runs[r].err; for (auto &q : runs) if (q.ctx) { fig_ctx_destroy(q.ctx); q.ctx = nullptr; } return true; } return false; };
    phase(shard_open);
    if (failed()) return 1;
    {   // the one exchange before the fill: every shard's reach bits, carried along the reference's worker processes
        std::vector<uint8_t> reach((size_t)std::max<int64_t>(ng, 1), 0);
        for (int r = 0; r < world; r++) for (size_t k = 0; k < runs[r].ids.size(); k++) reach[(size_t)runs[r].ids[k]] = runs[r].reach[k];
        ot_presets_from_reach(B, reach.data());
        for (int r = 0; r < world; r++) { runs[r].preset.assign(std::max<size_t>(runs[r].ids.size(), 1), 0); for (size_t k = 0; k < runs[r].ids.size(); k++) runs[r].preset[k] = B.gap_ot_preset[(size_t)runs[r].ids[k]]; }
    }
    phase(shard_fill);
    if (failed()) return 1;
    // ---- merge in global gap order (strings compacted as fig_fill_gaps leaves them) and global read order
    const int64_t NU = (int64_t)B.u_anchor_pos.size(), NP = (int64_t)B.p_pos.size();
    R.filled_len.assign(ng, 0); R.gaptofill.assign(ng, 0); R.str_off.assign(ng + 1, 0);
    R.draw_pos.assign((size_t)std::max<int64_t>(NU + NP, 1), INT32_MIN); R.draw_isz.assign((size_t)std::max<int64_t>(NU + NP, 1), 0); R.draw_len.assign((size_t)std::max<int64_t>(ng * 2, 1), -1);
    std::vector<int> owner((size_t)ng, -1); std::vector<int64_t> local((size_t)ng, 0);
    for (int r = 0; r < world; r++) for (size_t k = 0; k < runs[r].ids.size(); k++) { owner[runs[r].ids[k]] = r; local[runs[r].ids[k]] = (int64_t)k; }
    memset(&st, 0, sizeof(st));
    for (int r = 0; r < world; r++) {
        st.kernel_ms = std::max(st.kernel_ms, runs[r].st.kernel_ms); st.place_calls += runs[r].st.place_calls; st.alg_flops += runs[r].st.alg_flops;
        st.n_launches += runs[r].st.n_launches;
    }
    R.str.clear();
    for (int64_t g = 0; g < ng; g++) {
        const ShardRun &S = runs[owner[g]]; const int64_t k = local[g];
        R.filled_len[g] = S.R.filled_len[k]; R.gaptofill[g] = S.R.gaptofill[k];
        R.str_off[g] = (int64_t)R.str.size();
        R.str.append(S.R.str, (size_t)S.R.str_off[k], (size_t)(S.R.str_off[k + 1] - S.R.str_off[k]));
        R.draw_len[2 * g] = S.R.draw_len[2 * k]; R.draw_len[2 * g + 1] = S.R.draw_len[2 * k + 1];
        const int64_t nu = B.u_read_off[g + 1] - B.u_read_off[g], np = B.p_read_off[g + 1] - B.p_read_off[g];
        const int64_t snu = (int64_t)S.sub.u_anchor_pos.size();
        for (int64_t i = 0; i < nu; i++) { R.draw_pos[B.u_read_off[g] + i] = S.R.draw_pos[S.sub.u_read_off[k] + i]; R.draw_isz[B.u_read_off[g] + i] = S.R.draw_isz[S.sub.u_read_off[k] + i]; }
        for (int64_t i = 0; i < np; i++) { R.draw_pos[NU + B.p_read_off[g] + i] = S.R.draw_pos[snu + S.sub.p_read_off[k] + i]; R.draw_isz[NU + B.p_read_off[g] + i] = S.R.draw_isz[snu + S.sub.p_read_off[k] + i]; }
    }
    R.str_off[ng] = (int64_t)R.str.size();
    if (R.str.empty()) R.str.assign(1, 'N');
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 16) {
        fprintf(stderr, "usage: figfill <contigs.fa> <maxDistance> <readLen> <scriptItr> <partialFlag> <unmapped> <numThreads> "
                        "<myout.sam> <tmp/> <gaps/> <negOverlap> <partialReadLen> <trim> <setInputMean> <insertSize>\n");
        return 1;
    }
    RunArgs a;
    a.contigFile = argv[1]; a.D = atoi(argv[2]); a.read_length = atoi(argv[3]); a.script_itr = atoi(argv[4]);
    a.partial_flag = atoi(argv[5]); a.unmapped = atoi(argv[6]); a.num_threads = atoi(argv[7]); a.mapFile = argv[8];
    a.tmp = argv[9]; a.gapsDir = argv[10]; a.neg_overlap = atoi(argv[11]); a.partial_len = atoi(argv[12]);
    a.trim = atoi(argv[13]); a.setinputmean = atoi(argv[14]); a.isz = atoi(argv[15]);
    a.unm_limit = 400;                                   // gapthresh, FillGaps.cpp:22
    const char *dev_env = getenv("FIGFILL_DEVICE");
    int device = dev_env ? atoi(dev_env) : 0;
    std::vector<int> devices;                            // FIGFILL_DEVICES=0,1,...,7: one shard, host thread and fig_ctx per listed GPU
    if (const char *dl = getenv("FIGFILL_DEVICES")) {
        const char *q = dl;
        while (true) {
            char *e; long v = strtol(q, &e, 10);
            if (e == q || v < 0) { devices.clear(); break; }
            devices.push_back((int)v);
            if (*e == 0) break;
            if (*e != ',') { devices.clear(); break; }           // trailing garbage ("0,1x") is an error, not a shorter list
            q = e + 1;
        }
        if (devices.empty()) return fail("figfill: FIGFILL_DEVICES must be a comma-separated list of GPU ordinals");
    }
    setenv("GPU_MAX_HW_QUEUES", "8", 0);                 // one hardware queue per scheduler lane; before any thread or HIP call (fig_abi.hip: fig_ctx_create)
    auto t0 = std::chrono::steady_clock::now();

    std::string err;
    Scaffold sc;
    if (!load_scaffold(a.contigFile, sc, err)) return fail(err);
    Batch B;
    // FIGFILL_SAM="<maxD1>:<frag.sam>[;<maxD2>:<jump.sam>]": ingest the aligner's SAM in this process (the binning of
    // Preprocess.cpp, fig_sam.h) and hand the per-gap reads to the fill in memory -- no gaps_<g>.sam / partial_gaps_<g>.sam
    // files.  The run-level files (gapInfo/stat/stat2.txt, myout.sam) are still written: the model is built from them.
    if (const char *spec = getenv("FIGFILL_SAM")) {
        std::string sp(spec);
        std::vector<std::pair<int, std::string>> libs;
        size_t p = 0;
        while (p < sp.size()) { size_t e = sp.find(';', p); if (e == std::string::npos) e = sp.size(); std::string it = sp.substr(p, e - p); size_t c = it.find(':');
                                if (c != std::string::npos) libs.emplace_back(atoi(it.substr(0, c).c_str()), it.substr(c + 1)); p = e + 1; }
        if (libs.empty() || (a.unmapped == 1 && libs.size() < 2)) return fail("figfill: FIGFILL_SAM needs <maxD>:<frag.sam> and, in unmapped mode, ;<maxD>:<jump.sam>");
        figsam::Binned frag, jump;
        for (size_t k = 0; k < libs.size() && k < 2; k++) {
            figsam::Args pa;
            pa.contigFile = a.contigFile; pa.maxDistance = libs[k].first; pa.samflag = (int)k + 1; pa.mapFile = libs[k].second; pa.outFile = a.mapFile;
            pa.filledContigFile = a.contigFile; pa.reads1 = "r_1.fastq"; pa.reads2 = "r_2.fastq"; pa.gapsDir = a.gapsDir; pa.tmpDir = a.tmp; pa.default_setting = 1;
            if (k == 1 && a.unmapped != 1) break;
            if (figsam::preprocess(pa, k == 0 ? frag : jump, err, 2)) return fail(err);
        }
        if (!load_batch_mem(a, sc, a.unmapped == 1 ? &jump.gap_files : nullptr, &frag.partial_files, B, err)) return fail(err);
    } else if (!load_batch(a, sc, B, err)) return fail(err);
    printf("Total # of gaps = %zu\n", B.gap_contig.size());
    for (const std::string &m : B.messages) printf("%s\n", m.c_str());
    Model M;
    if (!build_model(a, sc, M, err)) return fail(err);

    fig_model fm; M.fill(fm, a);
    fig_gap_batch fb; B.view(fb, sc);
    if (devices.size() > 1) {
        Results R; fig_stats st;
        if (fill_multi(devices, a, sc, B, fm, R, st, err)) return fail(err);
        if (!write_gapout(a, B, R, err)) return fail(err);
        if (!write_draw(a, B, R, err)) return fail(err);
        if (!write_gaploads(a, B, err)) return fail(err);
        if (!write_scaffold(a, sc, B, R, err)) return fail(err);
        double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("Time taken = %g seconds (%zu GPUs; slowest shard's device kernels %.3f ms, %lld placeReads calls)\n", secs, devices.size(), st.kernel_ms, (long long)st.place_calls);
        printf("======================================\n");
        printf("Iteration %d ends successfully\n", a.script_itr);
        printf("======================================\n");
        return 0;
    }
    if (devices.size() == 1) device = devices[0];
    fig_ctx *ctx = nullptr;
    int rc = fig_ctx_create(device, &ctx);
    if (rc) return fail(std::string("figfill: fig_ctx_create: ") + fig_strerror(rc));
    rc = fig_ctx_set_model(ctx, &fm);
    if (rc) { fig_ctx_destroy(ctx); return fail(std::string("figfill: fig_ctx_set_model: ") + fig_strerror(rc)); }

    Results R;
    int64_t ng = fb.n_gaps;
    int64_t cap = fig_results_capacity(&fm, &fb);
    R.filled_len.assign(ng, 0); R.gaptofill.assign(ng, 0); R.str_off.assign(ng + 1, 0); R.str.assign((size_t)std::max<int64_t>(cap, 1), 'N');
    int64_t nr = (int64_t)B.u_anchor_pos.size() + (int64_t)B.p_pos.size();
    R.draw_pos.assign((size_t)std::max<int64_t>(nr, 1), INT32_MIN); R.draw_isz.assign((size_t)std::max<int64_t>(nr, 1), 0); R.draw_len.assign((size_t)std::max<int64_t>(ng * 2, 1), -1);
    fig_gap_results fr; memset(&fr, 0, sizeof(fr));
    fr.filled_len = R.filled_len.data(); fr.gaptofill = R.gaptofill.data(); fr.str_off = R.str_off.data();
    fr.str = &R.str[0]; fr.str_capacity = (int64_t)R.str.size();
    fr.draw_pos = R.draw_pos.data(); fr.draw_isz = R.draw_isz.data(); fr.draw_len = R.draw_len.data();
    // optional candidate trace for the parity tests (FIGFILL_TRACE=<file>)
    const char *trace = getenv("FIGFILL_TRACE");
    std::vector<int32_t> dn, di; std::vector<double> dl;
    const int maxc = 2048;
    if (trace) {
        dn.assign((size_t)std::max<int64_t>(ng, 1), 0); di.assign((size_t)std::max<int64_t>(ng, 1) * maxc * 3, 0); dl.assign((size_t)std::max<int64_t>(ng, 1) * maxc, 0);
        fr.dbg_max_cand = maxc; fr.dbg_n_cand = dn.data(); fr.dbg_cand_i = di.data(); fr.dbg_cand_lik = dl.data();
    }
    // upload; measure which gaps get to Figbird.cpp:6317; carry that along the reference's $num_threads worker processes
    // (overlap_threshold, :103); fill
    rc = fig_batch_upload(ctx, &fb);
    if (!rc && ng > 0) {
        std::vector<uint8_t> reach((size_t)ng, 0);
        rc = fig_batch_probe_reach(ctx, reach.data());
        if (!rc) { ot_presets_from_reach(B, reach.data()); rc = fig_batch_set_ot_preset(ctx, B.gap_ot_preset.data()); }
    }
    if (!rc) rc = fig_fill_resident(ctx, &fr);
    fig_stats st; memset(&st, 0, sizeof(st)); fig_get_stats(ctx, &st);
    fig_ctx_destroy(ctx);
    if (rc) return fail(std::string("figfill: fill: ") + fig_strerror(rc));

    if (!write_gapout(a, B, R, err)) return fail(err);
    if (!write_draw(a, B, R, err)) return fail(err);
    if (!write_gaploads(a, B, err)) return fail(err);
    if (!write_scaffold(a, sc, B, R, err)) return fail(err);
    if (trace) {
        FILE *f = fopen(trace, "w");
        if (f) {
            fprintf(f, "MODEL\t%d\t%d\t%d\t%a\t%a\t%a\n", M.cutoff, M.Tmin, M.Tmax, M.insertSizeMean, M.leftSD, M.rightSD);
            for (int64_t g = 0; g < ng; g++)
                for (int k = 0; k < dn[g] && k < maxc; k++)
                    fprintf(f, "CAND\t%d\t%d\t%d\t%a\t%d\n", (int)g, di[(g * maxc + k) * 3], di[(g * maxc + k) * 3 + 1], dl[g * maxc + k], di[(g * maxc + k) * 3 + 2]);
            fprintf(f, "STATS\t%lld\t%.17g\n", (long long)st.place_calls, st.alg_flops);
            fprintf(f, "LAUNCHES\t%d\n", (int)st.n_launches);      // kernel launches of the fill (the emulation reports its launch classes)
            fclose(f);
        }
    }
    double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("Time taken = %g seconds (device kernels %.3f ms, %lld placeReads calls)\n", secs, st.kernel_ms, (long long)st.place_calls);
    printf("======================================\n");
    printf("Iteration %d ends successfully\n", a.script_itr);
    printf("======================================\n");
    return 0;
}
