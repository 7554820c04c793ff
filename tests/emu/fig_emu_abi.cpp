// fig_emu_abi.cpp -- TEST INFRASTRUCTURE ONLY (CPU unit tests, `-m "not gpu"`).
//
// Implements the C ABI of include/figbird_hip.h by compiling the device engine
// (figbird_amd/csrc/fig_engine.h) for the host with FIG_EMU: ONE emulated lane per
// workgroup, barriers are no-ops.  It lets the CPU test-suite check the engine's control
// logic, packing and host plumbing against the oracle without a GPU.  It is never linked
// into libfighip.so or figfill; the product fails with FIG_ENODEV when no GPU is present.
#define FIG_EMU 1
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "../../include/figbird_hip.h"
#include "../../figbird_amd/csrc/fig_engine.h"
#include "../../figbird_amd/csrc/fig_pack.h"

struct fig_ctx {
    bool have_model = false;
    fig_model hm;
    FigDevModel dm;
    std::vector<double> e, ome, m3, insd, qtab, pairs;
    bool have_batch = false;
    FigPacked K;
    fig_stats stats;
};

extern "C" int fig_version(void) { return FIG_ABI_VERSION; }
extern "C" const char *fig_strerror(int code) {
    switch (code) {
        case FIG_OK: return "ok";
        case FIG_EINVAL: return "invalid argument";
        case FIG_ENODEV: return "no usable HIP device (libfighip has no CPU path)";
        case FIG_ENOMEM: return "out of memory";
        case FIG_EHIP: return "HIP runtime error";
        case FIG_ENOSPC: return "result string buffer too small";
        case FIG_EUNSUP: return "input outside the supported envelope";
        default: return "unknown error";
    }
}
extern "C" int fig_ctx_create(int, fig_ctx **out) { *out = new fig_ctx(); memset(&(*out)->stats, 0, sizeof(fig_stats)); return FIG_OK; }
extern "C" void fig_ctx_destroy(fig_ctx *c) { delete c; }
extern "C" void fig_batch_free(fig_ctx *c) { if (c) { c->K = FigPacked(); c->have_batch = false; } }
extern "C" int fig_get_stats(const fig_ctx *c, fig_stats *o) { *o = c->stats; return FIG_OK; }
extern "C" int64_t fig_results_capacity(const fig_model *m, const fig_gap_batch *b) { return fig_pack_results_capacity(m, b); }

extern "C" int fig_ctx_set_model(fig_ctx *ctx, const fig_model *m) {
    if (!ctx || !m) return FIG_EINVAL;
    if (m->max_read_length <= 0 || m->max_read_length > FIG_MAX_READLEN) return FIG_EUNSUP;
    if (m->partial_flag && m->unmapped_flag) return FIG_EUNSUP;
    fig_batch_free(ctx);                 // as libfighip: a resident batch was packed under the previous model
    int L = m->max_read_length;
    ctx->hm = *m;
    ctx->e.assign(m->error_pos_dist, m->error_pos_dist + L);
    ctx->ome.resize(L); ctx->m3.resize(L);
    for (int k = 0; k < L; k++) {
        volatile double a = 1 - m->error_pos_dist[k]; ctx->ome[k] = a;
        volatile double b = 1 - m->error_pos_dist[k] - m->in_pos_dist[k] - m->del_pos_dist[k]; ctx->m3[k] = b;
    }
    ctx->insd.assign(m->insert_len_dist_smoothed, m->insert_len_dist_smoothed + m->max_insert_size);
    ctx->insd.push_back(0.0);
    ctx->qtab.resize(256);
    for (int c = 0; c < 256; c++) { int Q = c - 33; ctx->qtab[c] = pow(10, -Q / 10.0); }
    FigDevModel &dm = ctx->dm;
    dm.L = L; dm.Tmin = m->insert_threshold_min; dm.Tmax = m->insert_threshold_max; dm.cutoff = m->gap_prob_cutoff;
    dm.partial_flag = m->partial_flag; dm.unmapped = m->unmapped_flag; dm.script_itr = m->script_itr; dm.D = m->max_distance;
    dm.read_length = m->read_length; dm.neg_overlap = m->neg_overlap; dm.partial_len = m->partial_len; dm.unm_limit = m->unm_limit;
    dm.max_insert = m->max_insert_size;
    for (int i = 0; i < 25; i++) dm.T[i] = m->error_type_probs[i];
    dm.fmm_up = fig_model_fmm(m);
    ctx->pairs.assign((size_t)8 * L, 0.0);
    for (int k = 0; k < L; k++) {
        ctx->pairs[2 * k] = ctx->ome[k]; ctx->pairs[2 * k + 1] = ctx->e[k];
        ctx->pairs[2 * L + 2 * k] = ctx->ome[L - 1 - k]; ctx->pairs[2 * L + 2 * k + 1] = ctx->e[L - 1 - k];
        ctx->pairs[4 * L + 2 * k] = ctx->m3[k]; ctx->pairs[4 * L + 2 * k + 1] = ctx->e[k];
        ctx->pairs[6 * L + 2 * k] = ctx->m3[L - 1 - k]; ctx->pairs[6 * L + 2 * k + 1] = ctx->e[L - 1 - k];
    }
    dm.e = ctx->e.data(); dm.ome = ctx->pairs.data(); dm.ome1 = ctx->ome.data(); dm.m3 = ctx->m3.data(); dm.insd = ctx->insd.data(); dm.qtab = ctx->qtab.data();
    ctx->have_model = true;
    return FIG_OK;
}

extern "C" int fig_batch_upload(fig_ctx *ctx, const fig_gap_batch *b) {
    if (!ctx || !b || !ctx->have_model) return FIG_EINVAL;
    ctx->K = FigPacked();
    int rc = fig_pack(&ctx->hm, b, sizeof(FigState), ctx->K);
    if (rc) return rc;
    ctx->have_batch = true;
    const int64_t ng = ctx->K.n_gaps;
    if (!ctx->K.ot_given && ctx->hm.partial_flag && ng > 0) {      // as fig_abi.hip: one process in batch order, carry measured
        std::vector<uint8_t> reach((size_t)ng, 0), preset((size_t)ng, 0);
        if ((rc = fig_batch_probe_reach(ctx, reach.data()))) return rc;
        std::vector<int64_t> ids((size_t)ng);
        for (int64_t g = 0; g < ng; g++) ids[(size_t)g] = g;
        fig_ot_carry(ids, reach.data(), preset.data());
        if ((rc = fig_batch_set_ot_preset(ctx, preset.data()))) return rc;
    }
    return FIG_OK;
}

static int emu_run(fig_ctx *ctx, fig_gap_results *out, uint8_t *probe_reach);

extern "C" int fig_fill_resident(fig_ctx *ctx, fig_gap_results *out) {
    if (!ctx || !out || !ctx->have_batch) return FIG_EINVAL;
    return emu_run(ctx, out, nullptr);
}

// fig_batch_probe_reach / fig_batch_set_ot_preset as in fig_abi.hip
extern "C" int fig_batch_probe_reach(fig_ctx *ctx, uint8_t *reach) {
    if (!ctx || !reach || !ctx->have_batch) return FIG_EINVAL;
    memset(reach, 0, (size_t)ctx->K.n_gaps);
    if (!ctx->hm.partial_flag || ctx->K.n_gaps == 0) return FIG_OK;
    return emu_run(ctx, nullptr, reach);
}
extern "C" int fig_batch_set_ot_preset(fig_ctx *ctx, const uint8_t *preset) {
    if (!ctx || !preset || !ctx->have_batch) return FIG_EINVAL;
    for (int64_t g = 0; g < ctx->K.n_gaps; g++) ctx->K.ot_preset[(size_t)g] = preset[g] ? 1 : 0;
    return FIG_OK;
}

// out != nullptr: the fill; probe_reach != nullptr: the reach pre-pass (fig_probe_kernel's role)
static int emu_run(fig_ctx *ctx, fig_gap_results *out, uint8_t *probe_reach) {
    fig_gap_results none; memset(&none, 0, sizeof(none));
    if (!out) out = &none;
    FigPacked &K = ctx->K;
    int64_t ng = K.n_gaps;
    FigDevBatch B; memset(&B, 0, sizeof(B));
    B.n_gaps = ng; B.gaps = K.gaps.data(); B.order = K.order.data();
    B.u.pos = K.u_pos.data(); B.u.aux = K.u_aux.data(); B.u.len = K.u_len.data(); B.u.woff = K.u_woff.data();
    B.p.pos = K.p_pos.data(); B.p.aux = K.p_aux.data(); B.p.clip = K.p_clip.data(); B.p.refpos = K.p_ref.data();
    B.p.len = K.p_len.data(); B.p.woff = K.p_woff.data(); B.p.qoff = K.p_qoff.data();
    B.packed = K.packed.data(); B.qual = K.qual.data(); B.flank = K.flank.data();
    std::vector<int32_t> fl(ng + 1, 0), gtf(ng + 1, 0);
    std::vector<char> str((size_t)K.str_total + 8, 'N');
    B.filled_len = fl.data(); B.gaptofill = gtf.data(); B.str = str.data();
    if (out->dbg_n_cand && out->dbg_cand_i && out->dbg_cand_lik && out->dbg_max_cand > 0) {
        B.dbg_max_cand = out->dbg_max_cand; B.dbg_n_cand = out->dbg_n_cand; B.dbg_cand_i = out->dbg_cand_i; B.dbg_cand_lik = out->dbg_cand_lik;
        for (int64_t g = 0; g < ng; g++) out->dbg_n_cand[g] = 0;
        B.dbg_n_place = out->dbg_n_place;
        if (out->dbg_counts && out->dbg_plane_cols > 0) { B.dbg_counts = out->dbg_counts; B.dbg_plane_cols = out->dbg_plane_cols; memset(out->dbg_counts, 0, (size_t)ng * out->dbg_max_cand * out->dbg_plane_cols * 5 * 8); }
        if (out->dbg_read_maxlv && out->dbg_plane_reads > 0) { B.dbg_read_maxlv = out->dbg_read_maxlv; B.dbg_plane_reads = out->dbg_plane_reads; memset(out->dbg_read_maxlv, 0, (size_t)ng * out->dbg_max_cand * out->dbg_plane_reads * 8); }
    }
    if (out->draw_pos && out->draw_isz && out->draw_len) { B.draw_pos = out->draw_pos; B.draw_isz = out->draw_isz; B.draw_len = out->draw_len; }
    B.n_ureads = (int64_t)K.u_pos.size();
    int32_t qh = 0; unsigned long long counters[8] = {0};
    B.queue_head = &qh; B.counters = counters;
    long long stride = fig_scratch_layout(nullptr, K.capG, K.capR, K.capP, K.capC, K.capW, K.capE, nullptr);
    std::vector<unsigned char> slab((size_t)stride + 64, 0);
    B.scratch = slab.data(); B.scratch_stride = stride;
    B.capG = K.capG; B.capR = K.capR; B.capP = K.capP; B.capC = K.capC; B.capW = K.capW; B.capE = K.capE;
    std::vector<unsigned char> persist((size_t)K.persist_total + 256, 0);
    B.persist = persist.data();
    std::vector<int32_t> gapctl((size_t)ng * 4 + 4, 0);
    B.gapctl = gapctl.data();
    B.ot_preset = K.ot_preset.data();
    const FigDevModel &M = ctx->dm;
    for (const FigLaunchClass &c : K.classes) {
        // one emulated lane = one wave of width 1; the class's team count is kept so the chunking logic runs
        int nteams = 1;
        // LDS-tiled form of a class (fig_pack.h): what the packer chose, or forced on every class by FIG_EMU_TILES=<n> so that the
        // tile logic of the E-step is exercised on small gaps too
        int tiles = c.tiles, tile_step = c.tile_step, tile_cols = c.tile_cols;
        if (const char *ft = getenv("FIG_EMU_TILES")) {
            tiles = atoi(ft);
            if (tiles > 0) { tile_step = ((c.ncolE + tiles - 1) / tiles + 7) & ~7; tile_cols = (tile_step + M.L + 8 + 7) & ~7; }
            fprintf(stderr, "[figemu] class ncolE=%d: LDS-tiled E-step forced, tiles=%d step=%d cols=%d\n", c.ncolE, tiles, tile_step, tile_cols);
        }
        const long long tabc = tiles > 0 && tile_cols > c.ncolE ? tile_cols : c.ncolE;
        std::vector<double> lds((size_t)(9LL * tabc + (long long)nteams * c.Wcap) + (sizeof(FigState) + c.capGl + FIG_MAX_READLEN + 64 + FIG_PLB_BYTES) / 8 + 8, 0.0);
        fig_lds = lds.data();
        FigEng E;
        E.tid = 0; E.nt = 1; E.lane = 0; E.wave = 0; E.nw = 1; E.wsz = 1;
        E.M = &ctx->dm; E.B = &B; E.capG = c.capG; E.flops = 0; E.mle_alg = 0; E.mle_exec = 0;
        E.ncolE = c.ncolE; E.xoff = M.L - 1; E.Wcap = c.Wcap; E.nteams = nteams;
        fig_scratch_layout(slab.data(), K.capG, K.capR, K.capP, K.capC, K.capW, K.capE, &E.scr);
        E.pq_lds = 1; E.w_lds = 1;
        E.off_pq = 0; E.off_q4 = 8 * c.ncolE; E.off_w = 9 * c.ncolE;
        E.pq = (FigPQ *)(fig_lds + E.off_pq); E.q4 = fig_lds + E.off_q4; E.wbuf = fig_lds + E.off_w;
        E.tiles = tiles; E.tile_step = tile_step; E.tile_cols = tile_cols;
        if (tiles > 0) {      // as fig_eng_init does: image of one tile + the weight rows in "LDS", table and MLE buffers in the slab
            E.pq_lds = 0;
            E.off_q4 = 8 * tile_cols; E.off_w = 9 * tile_cols;
            E.pq = E.scr.pqg; E.q4 = E.scr.q4g; E.wbuf = E.scr.wg;
        }
        E.S = (FigState *)(fig_lds + 9LL * tabc + (long long)nteams * c.Wcap); E.lds_tw = (int)(9LL * tabc + (long long)nteams * c.Wcap);
        unsigned char *bp = (unsigned char *)(E.S + 1);
        E.gs = bp; bp += ((c.capGl + 7) & ~7); E.rb = bp; bp += ((FIG_MAX_READLEN + 8 + 15) & ~15); E.plb = (uint32_t *)bp; E.off_plb = 0;
        E.kt_fwd = M.ome; E.kt_rev = M.ome + 2 * M.L; E.mt_fwd = M.ome + 4 * M.L; E.mt_rev = M.ome + 6 * M.L;
        FigScr work = E.scr;
        const char *sched = getenv("FIG_SCHED");
        const bool seq = sched && strcmp(sched, "seq") == 0;
        auto persist_of = [&](const FigDevGap &g, FigPersist &P) { fig_persist_layout(B.persist + g.persistOff, g.capGg, g.nU, g.nP, g.rangeCap, g.nslots, sizeof(FigState), &P); };
        if (probe_reach) {
            for (int qi = c.q_begin; qi < c.q_end; qi++) {
                E.g = &K.gaps[K.order[qi]];
                FigPersist P; persist_of(*E.g, P);
                fig_bind(E, work, P, FIG_BIND_SEQ, 0, E.g->capGg);
                probe_reach[K.order[qi]] = fig_gap_probe<true>(E) ? 1 : 0;
            }
        } else if (seq) {
            for (int qi = c.q_begin; qi < c.q_end; qi++) {
                E.g = &K.gaps[K.order[qi]];
                FigPersist P; persist_of(*E.g, P);
                fig_bind(E, work, P, FIG_BIND_SEQ, 0, E.g->capGg);
                fig_fill_gap<true>(E);
            }
        } else {
            // candidate-parallel schedule, executed in order (mirrors run_class_parallel in fig_abi.hip)
            std::vector<int> ids(K.order.begin() + c.q_begin, K.order.begin() + c.q_end);
            std::vector<int> status(K.gaps.size(), 0);
            for (int gi : ids) {
                E.g = &K.gaps[gi];
                FigPersist P; persist_of(*E.g, P);
                fig_bind(E, work, P, FIG_BIND_SEQ, 0, E.g->capGg);
                fig_gap_begin<true>(E);
                if (!E.S->L.inr) { while (!E.S->L.done) { fig_eval_candidate<true>(E); fig_loop_step(E); } fig_gap_end<true>(E); status[gi] = 0; }
                else { fig_state_save(E, P); status[gi] = E.S->L.done ? 2 : 1; }
            }
            const char *ce = getenv("FIG_EMU_CHUNK");
            const int chunk = ce ? atoi(ce) : 5;
            while (true) {
                int n_active = 0;
                for (int gi : ids) n_active += status[gi] == 1;
                if (!n_active) break;
                for (int gi : ids) {
                    if (status[gi] != 1) continue;
                    E.g = &K.gaps[gi];
                    FigPersist P; persist_of(*E.g, P);
                    const FigState *snap = (const FigState *)P.state;
                    int j = snap->L.j, range = snap->L.range;
                    int n = std::min(std::min(chunk, E.g->nslots), range - j);
                    for (int k = n - 1; k >= 0; k--) {                      // any order; every item starts from poisoned scratch + LDS
                        memset(slab.data(), 0xA5, slab.size()); memset(lds.data(), 0xA5, lds.size() * sizeof(double));
                        fig_spec_eval<true>(E, work, P, j + k, k, E.g->capGg);
                    }
                    memset(slab.data(), 0xA5, slab.size()); memset(lds.data(), 0xA5, lds.size() * sizeof(double));
                    fig_spec_replay(E, work, P, n, E.g->capGg);
                    status[gi] = E.S->L.done ? 2 : 1;
                }
            }
            for (int gi : ids) {
                if (status[gi] != 2) continue;
                E.g = &K.gaps[gi];
                FigPersist P; persist_of(*E.g, P);
                memset(slab.data(), 0xA5, slab.size()); memset(lds.data(), 0xA5, lds.size() * sizeof(double));
                fig_bind(E, work, P, FIG_BIND_SEQ, 0, E.g->capGg);
                fig_state_load(E, P);
                fig_gap_end<true>(E);
            }
        }
        counters[1] += E.flops; counters[3] += E.mle_alg; counters[4] += E.mle_exec;
    }
    if (probe_reach) return FIG_OK;
    ctx->stats.place_calls = (int64_t)counters[0]; ctx->stats.alg_flops = (double)counters[1];
    ctx->stats.spec_flops = (double)counters[1]; ctx->stats.mle_alg_flops = (double)counters[3]; ctx->stats.mle_exec_flops = (double)counters[4];
    ctx->stats.packed_bytes = K.packed_bytes(); ctx->stats.n_launches = (int)K.classes.size();
    int64_t need = 0;
    for (int64_t g = 0; g < ng; g++) { out->filled_len[g] = fl[g]; out->gaptofill[g] = gtf[g]; need += fl[g] > 0 ? fl[g] : 0; }
    if (need > out->str_capacity) return FIG_ENOSPC;
    int64_t o = 0;
    for (int64_t g = 0; g < ng; g++) {
        out->str_off[g] = o;
        if (fl[g] > 0) { memcpy(out->str + o, str.data() + K.str_off[g], (size_t)fl[g]); o += fl[g]; }
    }
    out->str_off[ng] = o;
    return FIG_OK;
}

extern "C" int fig_fill_gaps(fig_ctx *ctx, const fig_gap_batch *batch, fig_gap_results *out) {
    int rc = fig_batch_upload(ctx, batch);
    if (rc) return rc;
    rc = fig_fill_resident(ctx, out);
    fig_batch_free(ctx);
    return rc;
}
