// fig_engine_sched.h -- fillGap (Figbird.cpp:6201-6570, with analyzeGap :6168-6199 and findGapLeftRight
// :2151-2174) cut into begin / evaluate-candidate / step / end.  See FigLoop in fig_engine.h.
#ifndef FIG_ENGINE_SCHED_H
#define FIG_ENGINE_SCHED_H

// ---- allocate + analyzeGap + checkGapReads (:1638-1778, :7342-7343, :6208-6294)
template <bool LDS>
FIG_D void fig_gap_begin(FigEng &E) {
    FigState &S = *E.S;
    const FigDevModel &M = *E.M;
    const FigDevGap &g = *E.g;
    int G0 = g.G0, nU = g.nU;
    if (E.tid == 0) {
        S.left = M.D; S.right = M.D; S.side_limit = 30; S.end_pos_max = 0;
        S.valid_count = S.invalid_count = 0;
        S.partial_read_count = M.partial_flag == 1 ? g.nP : 0;       // r_count2 (findcount_file(...,1)) in partial mode
        S.region_perct = 0; S.region_perct_max = 0;
        S.partial_read_len = M.partial_len;
        S.rep_flag = 0; S.one_side_repeat_flag = 0; S.large_gap_flag = g.lgf; S.comp_count = 0;
        S.left_max = 0; S.right_min = 0; S.discont_or_not = 0;
        S.psr_temp[0] = S.psr_temp[1] = S.psr_final[0] = S.psr_final[1] = -1;
        S.umaxleftf = S.umaxrightf = S.ucoverf = 0;
        // overlap_threshold is a file-level global in the reference (Figbird.cpp:103), 0 until the first candidate loop of the
        // process reaches :6317 and 5 ever after: a gap sees 5 if an EARLIER gap of its worker process got there (ot_preset:
        // the prefix-OR, in the process's gap order, of the bits fig_gap_probe measures) or once its own loop does
        // (fig_eval_candidate / fig_spec_replay)
        S.num_itr = 0; S.overlap_threshold = (E.B->ot_preset && E.B->ot_preset[g.gapNo]) ? 5 : 0; S.gaptofill = 0;
        S.gl_len = S.gr_len = S.pl_len = S.pr_len = 0;
        S.cons_len = 1; E.scr.cons[0] = 4;
        // used_read_arr is an uninitialised stack array in the reference (Figbird.cpp:6265); the one element it can read before
        // writing is [0], when the loop leaves at its first candidate by side_limit < 10 in partial mode (:6303 -> :6547).  The
        // reference's own builds disagree there (-O0: the stack residue is nonzero in ~70 % of sampled gaps, zero otherwise);
        // the oracle and the engine take 0, i.e. the run(originalGap) + finalize(originalGap) branch
        E.scr.used_read_arr[0] = 0;
        S.best_len = S.cur_len = S.prev_len = S.orig_len = 0;
        S.flops_useful = 0; S.n_place = 0; S.dbg_j = -1; S.pad_dj = 0;
        fig_flank_tables(E);
        if (E.B->dbg_n_cand) E.B->dbg_n_cand[g.gapNo] = 0;
        if (E.B->draw_len) { E.B->draw_len[(long long)g.gapNo * 2] = -1; E.B->draw_len[(long long)g.gapNo * 2 + 1] = -1; }
    }
    for (int r = E.tid; r < nU; r += E.nt) {
        E.scr.saved[r] = 0; E.scr.mark[r] = 0; E.scr.org[r * 2] = -200; E.scr.org[r * 2 + 1] = 0;
        E.scr.frp[r * 2] = -200; E.scr.frp[r * 2 + 1] = 0;
        if (E.B->draw_pos) E.B->draw_pos[g.uBase + r] = (int)0x80000000;
    }
    for (int p = E.tid; p < g.nP; p += E.nt) {
        E.scr.ppos_org[p * 3] = 0; E.scr.ppos_org[p * 3 + 1] = -200; E.scr.ppos_org[p * 3 + 2] = 0;
        if (E.B->draw_pos) E.B->draw_pos[E.B->n_ureads + g.pBase + p] = (int)0x80000000;
    }
    FIG_SYNC();
    // ---- analyzeGap: findGapLeftRight + findRepeat
    if (E.tid == 0) {
        S.G = G0; fig_ise(E);
        int n = 0;
        for (int k = S.side_limit; k >= 1; k--) S.gap_left[n++] = (unsigned char)fig_flank_l(E, k);   // last side_limit bases before the gap
        S.gl_len = n;
        n = 0;
        for (int k = 0; k < S.right && n < S.side_limit; k++) {
            long long abs = g.gapStart + G0 + k;
            S.gap_right[n++] = (unsigned char)(abs < g.contigLen ? fig_flank_r(E, k) : 255);
        }
        S.gr_len = n;
    }
    FIG_SYNC();
    fig_find_repeat(E);
    if (E.tid == 0) {
        FigLoop &L = S.L;
        float gp_frac1 = g.gpf1, gp_frac2 = g.gpf2;
        int skip = 0;
        if (S.rep_flag == 1 && M.partial_flag) skip = 1;
        else if (S.one_side_repeat_flag == 1 && M.partial_flag && G0 > 3 * (2 * M.partial_len)) skip = 1;
        else if (g.fillflag == -1) skip = 1;
        else if (S.partial_read_count == 0 && nU == 0) skip = 1;
        int inr = !skip;                                 // g in [0, 765466] / [0, 965757] otherwise (:6194-6196)
        if (inr && g.gapNo > (M.script_itr == 1 ? 765466 : 965757)) inr = 0;
        S.num_itr = inr ? 200 : 0;
        if (!inr) { gp_frac1 = 1; gp_frac2 = 1; }
        int gapMin = (int)(G0 * gp_frac1), gapMax = (int)(G0 * gp_frac2);
        L.inr = inr; L.gapMin = gapMin; L.gapEstimate = gapMin; L.maxGapEstimate = gapMin; L.j = 0;
        L.maxLikelihood = -FIG_DBL_MAX; L.secondMaxLikelihood = -FIG_DBL_MAX; L.likelihood = 0; L.prevlikelihood = 0;
        L.fill_or_not = 0; L.same_count = 0; L.same_thresh = M.unmapped ? 50 : 4; L.stuckCount = 0;
        int range = gapMax - gapMin + 1;
        if (range > E.B->capC) range = E.B->capC;        // host sizes capC from the same formula; defensive
        L.range = range; L.side_flag = 0; L.less_read_flag = 0;
        L.prev_best = -1; L.curr_best = 0; L.prev_u = -1; L.curr_u = 0; L.sec_same = 0; L.sec_same2 = 0;
        L.done = 0; L.need_orig = 0; L.ev_iters = 0; L.ev_side_break = 0; L.ev_fill = 0;
        L.finalize_flag = 1;
        if (M.unmapped && S.large_gap_flag == 0) L.finalize_flag = 0;
    }
    FIG_SYNC();
    if (M.unmapped && G0 <= M.unm_limit && S.L.inr) {
        int lrf = fig_check_gap_reads<LDS>(E, G0);
        if (E.tid == 0) {
            S.L.less_read_flag = lrf;
            if (lrf == 1) S.L.range = 0;
            if (lrf == -2) { S.L.side_flag = 1; S.L.range = 0; }
        }
    }
    if (E.tid == 0 && S.L.range <= 0) S.L.done = 1;
    FIG_SYNC();
}

// ---- does this gap's candidate loop get to Figbird.cpp:6317 (`overlap_threshold=5`)?  The loop leaves before that line only
// at its first candidate: no candidate at all (checkGapReads, :6292-6293), side_limit < 10 after the first initialize()
// (:6303) or the gap closed by a negative overlap (:6305-6306); a later candidate is only reached past :6317.  None of this
// reads overlap_threshold, so the bit can be measured before any gap is filled.  Partial mode only: the threshold is read
// by detect_overlap_gapestimate alone (:2684, :2760-2766), whose callers are the partial placeReads (:3501) and the
// `if(partial_flag==1)` block of finalize (:5512-5574), so an unmapped-mode run never looks at it.
template <bool LDS>
FIG_D int fig_gap_probe(FigEng &E) {
    fig_gap_begin<LDS>(E);
    FigState &S = *E.S;
    if (S.L.range <= 0) return 0;
    int fill = fig_initialize(E, S.L.gapEstimate, 0);
    if (S.side_limit < 10) return 0;
    if (S.one_side_repeat_flag == 1) fill = 0;
    return !(fill != 0 && S.L.inr);
}

// ---- one candidate length: initialize + EM loop (+ the extra finalize-flag placeReads) + consensus (:6300-6356).
// Inputs: S.L.gapEstimate, S.L.j and the window state in S.  Outputs: S.L.ev_*, S.L.likelihood, S.valid_count,
// S.region_perct, S.umax*/ucoverf, S.discont_or_not, S.psr_temp, E.scr.cons (S.cons_len), E.scr.mark, and
// (when gapEstimate == originalGap) E.scr.org / E.scr.ppos_org.
template <bool LDS>
FIG_D void fig_eval_candidate(FigEng &E) {
    FigState &S = *E.S;
    const FigDevModel &M = *E.M;
    const int G0 = E.g->G0;
    const int gapEstimate = S.L.gapEstimate, j = S.L.j, inr = S.L.inr, finalize_flag = S.L.finalize_flag;
    FIG_SYNC();
    FIG_T0(E);
    if (E.tid == 0) { S.umaxleftf = S.umaxrightf = S.ucoverf = 0; S.L.ev_side_break = 0; S.L.ev_iters = 0; S.dbg_j = j; }
    int fill = fig_initialize(E, gapEstimate, j);
    FIG_TICK(E, 35);
    if (S.side_limit < 10) { if (E.tid == 0) { S.L.ev_side_break = 1; S.L.ev_fill = fill; S.dbg_j = -1; } FIG_SYNC(); return; }
    if (S.one_side_repeat_flag == 1) fill = 0;
    if (E.tid == 0) S.L.ev_fill = fill;
    if (fill != 0 && inr) { if (E.tid == 0) S.dbg_j = -1; FIG_SYNC(); return; }
    int i = 0;
    if (E.tid == 0) { S.discont_or_not = 0; S.comp_count = 0; S.overlap_threshold = 5; S.lik = S.L.likelihood; }
    FIG_SYNC();
    int preset_unfilled_len = 2 * M.read_length;
    for (i = 0; i < S.num_itr; i++) {
        if (E.tid == 0) { S.valid_count = 0; S.invalid_count = 0; }
        FIG_SYNC();
        fig_place_reads<LDS>(E, i, finalize_flag, gapEstimate - G0, S.large_gap_flag);
        FIG_TICK(E, 38);
        fig_compute_probs(E);
        FIG_SYNC();
        FIG_TICK(E, 36);
        if (M.unmapped) {
            if (S.comp_count >= 5) break;
            if (S.large_gap_flag == 1 && S.region_perct * gapEstimate < preset_unfilled_len) break;
        } else { if (i == 2) break; }
    }
    if (M.unmapped && !finalize_flag && inr) {
        if (E.tid == 0) S.valid_count = 0;
        FIG_SYNC();
        fig_place_reads<LDS>(E, i, 1, gapEstimate - G0, 0);
    }
    fig_compute_sequence(E, 0, 0);
    if (E.tid == 0) { S.L.likelihood = S.lik; S.L.ev_iters = i; S.dbg_j = -1; }
    FIG_SYNC();
    FIG_TICK(E, 37);
}

// ---- bookkeeping after a candidate (:6358-6480): best tracking, early-stop heuristics.  Consumes the outputs
// of fig_eval_candidate (wherever they were computed) and advances S.L (j, gapEstimate, done, need_orig).
FIG_D void fig_loop_step(FigEng &E) {
    FigState &S = *E.S;
    const FigDevModel &M = *E.M;
    const int G0 = E.g->G0, nU = E.g->nU;
    FIG_SYNC();
    if (S.L.ev_side_break) { if (E.tid == 0) { S.L.side_flag = 1; S.L.fill_or_not = S.L.ev_fill; S.L.done = 1; } FIG_SYNC(); return; }
    if (S.L.ev_fill != 0 && S.L.inr) { if (E.tid == 0) { S.L.fill_or_not = S.L.ev_fill; S.L.done = 1; } FIG_SYNC(); return; }
    const int j = S.L.j, gapEstimate = S.L.gapEstimate;
    const double likelihood = S.L.likelihood;
    const int valid_count = S.valid_count;
    const bool newbest = likelihood > S.L.maxLikelihood;
    if (E.tid == 0) fig_dbg_cand(E, gapEstimate, S.L.ev_iters, valid_count, likelihood);
    if (newbest) {
        fig_copy_str(E, E.scr.best, &S.best_len, E.scr.cons, S.cons_len);
        for (int k = E.tid; k < nU; k += E.nt) E.scr.saved[k] = E.scr.mark[k];
    }
    if (gapEstimate == G0) fig_copy_str(E, E.scr.orig, &S.orig_len, E.scr.cons, S.cons_len);
    FIG_SYNC();
    if (E.tid == 0) {
        FigLoop &L = S.L;
        L.fill_or_not = L.ev_fill;
        if (newbest) {
            L.secondMaxLikelihood = L.maxLikelihood;
            L.maxLikelihood = likelihood;
            L.maxGapEstimate = gapEstimate;
            S.region_perct_max = S.region_perct; S.psr_final[0] = S.psr_temp[0]; S.psr_final[1] = S.psr_temp[1];
            L.curr_best = j;
            L.prev_u = valid_count;
        } else if (likelihood > L.secondMaxLikelihood) L.secondMaxLikelihood = likelihood;
        if (j < E.B->capC) { E.scr.used_read_arr[j] = valid_count; E.scr.lrmd[j * 2] = S.left; E.scr.lrmd[j * 2 + 1] = S.right; }
        int brk = 0, need = 0;
        if (M.partial_flag || M.unmapped) {
            double diff1 = L.prevlikelihood - likelihood; if (diff1 < 0) diff1 = -diff1;
            if (diff1 <= 0.9) L.same_count++; else L.same_count = 0;
            L.prevlikelihood = likelihood;
            const int st = L.same_thresh;
            if (L.same_count == st) { brk = 1; need = S.G < G0; }
            if (!brk && M.unmapped) {
                L.curr_u = valid_count;
                int du = L.curr_u - L.prev_u; if (du < 0) du = -du;
                if (L.curr_best == L.prev_best && du <= 2) L.sec_same++;
                else { L.prev_best = L.curr_best; L.sec_same = 0; }
                if (L.sec_same >= 2 * st) { brk = 1; need = S.G < G0; }
                if (!brk && G0 <= 30) {
                    if (!(S.umaxleftf == 1 || S.umaxrightf == 1 || S.ucoverf == 1)) L.sec_same2++; else L.sec_same2 = 0;
                    if (L.sec_same2 >= 1.5 * st) { brk = 1; need = S.G < G0; }
                }
                if (!brk) {
                    if (S.discont_or_not == 1 && valid_count < 5) L.stuckCount++; else L.stuckCount = 0;
                    if (L.stuckCount > 3 * st) { brk = 1; need = S.G < G0; }
                }
            }
        }
        if (brk) { L.done = 1; L.need_orig = need; }
        else { L.gapEstimate++; L.j++; if (L.j >= L.range) L.done = 1; }
    }
    FIG_SYNC();
}

// ---- after the loop (:6484-6569) + the gapout record (:7411-7413)
template <bool LDS>
FIG_D void fig_gap_end(FigEng &E) {
    FigState &S = *E.S;
    const FigDevModel &M = *E.M;
    const FigDevGap &g = *E.g;
    const int G0 = g.G0;
    FIG_SYNC();
    if (S.L.need_orig) fig_run_original<LDS>(E, E.scr.orig, &S.orig_len);   // the run(originalGap) of the break paths (:6411-6413 ...)
    const FigLoop L = S.L;
    FIG_SYNC();
    if (L.fill_or_not != 0 && L.inr) {
        if (E.tid == 0) { S.G = 0; S.gaptofill = L.fill_or_not; }
    } else if (L.inr) {
        if (M.unmapped) {
            if (L.less_read_flag == 1) {
                fig_run_original<LDS>(E, E.scr.orig, &S.orig_len);
                fig_finalize<LDS>(E, G0);
            } else if (L.side_flag) {
                fig_run_original<LDS>(E, E.scr.best, &S.best_len);
                fig_finalize<LDS>(E, G0);
            } else {
                int changed = 0;                         // check_change(used_read_arr, j), :5886-5895
                if (L.j != 1) for (int i = 1; i < L.j && i < E.B->capC; i++) if (E.scr.used_read_arr[0] != E.scr.used_read_arr[i]) { changed = 1; break; }
                if (changed) fig_finalize<LDS>(E, L.maxGapEstimate);
                else { fig_copy_str(E, E.scr.best, &S.best_len, E.scr.orig, S.orig_len); fig_finalize<LDS>(E, G0); }
            }
        } else {
            if (L.maxGapEstimate == 0) {
                if (E.scr.used_read_arr[0] != 0) fig_finalize<LDS>(E, L.maxGapEstimate);
                else {
                    if (L.maxGapEstimate < G0) fig_run_original<LDS>(E, E.scr.orig, &S.orig_len);
                    fig_finalize<LDS>(E, G0);
                }
            } else {
                if (L.side_flag) {
                    int idx = L.maxGapEstimate - L.gapMin;
                    if (E.tid == 0 && idx >= 0 && idx < E.B->capC) { S.left = E.scr.lrmd[idx * 2]; S.right = E.scr.lrmd[idx * 2 + 1]; }
                    FIG_SYNC();
                }
                fig_finalize<LDS>(E, L.maxGapEstimate);
            }
        }
    }
    FIG_SYNC();
    int len = S.G;
    const char A[5] = {'A', 'C', 'G', 'T', 'N'};
    char *dst = E.B->str + g.strOff;
    for (int x = E.tid; x < len && x < g.alloc_arg; x += E.nt) {
        int c = x < S.cons_len ? E.scr.cons[x] : 4;
        dst[x] = A[c > 4 ? 4 : c];
    }
    if (E.tid == 0) {
        E.B->filled_len[g.gapNo] = len; E.B->gaptofill[g.gapNo] = S.gaptofill;
        fig_atomic_add_u64(&E.B->counters[0], (unsigned long long)S.n_place);
        if (E.B->dbg_n_place) E.B->dbg_n_place[g.gapNo] = S.n_place;
        if (S.flops_useful) fig_atomic_add_u64(&E.B->counters[1], S.flops_useful);
    }
    FIG_SYNC();
}

// Sequential mode: the whole gap in one workgroup.
template <bool LDS>
FIG_D void fig_fill_gap(FigEng &E) {
    fig_gap_begin<LDS>(E);
    while (!E.S->L.done) {
        fig_eval_candidate<LDS>(E);
        fig_loop_step(E);
    }
    fig_gap_end<LDS>(E);
}


// =======================================================================================
// Candidate-parallel mode.  The candidate lengths of one gap are (almost) independent EM runs -- what one
// candidate leaves for the next is only monotone window state and strings of a different length -- so they are
// evaluated speculatively in different workgroups, CHUNK at a time, and the sequential bookkeeping
// (fig_loop_step) is replayed in candidate order by one small workgroup per gap.  Candidates evaluated past an
// early stop are discarded (<= CHUNK-1 per gap).  Per-gap state lives in a persistent slab in HBM:
//
//   [FigState snapshot][best][orig][prev][saved][org][ppos_org][repeatflag][used_read_arr][lrmd]
//   [org_spec][ppos_spec]                      <- written by the candidate whose length == originalGap
//   [slot 0 .. nslots-1]: {FigSlot header, consensus[capGg+8], previous_str[capGg+8], mark_accepted[nU+8]}
struct FigSlot {
    double lik, region_perct;
    long long end_pos_max;
    unsigned long long flops;
    int G, iters, valid, side_break, fill, flags, psr0, psr1, left, right, side_limit, cons_len, partial_read_len, partial_read_count, n_place, prev_len;
};

struct FigPersist {                  // pointers into one gap's persistent slab
    unsigned char *state;            // FigState snapshot
    unsigned char *best, *orig, *prev, *saved;
    int *org, *ppos_org, *repeatflag, *used_read_arr, *lrmd, *org_spec, *ppos_spec;
    unsigned char *slots; long long slot_stride;
};

FIG_HD long long fig_persist_layout(unsigned char *base, int capGg, int nU, int nP, int range, int nslots, size_t state_bytes, FigPersist *p) {
    long long o = 0;
#define FIG_PC(field, type, count) do { if (p) p->field = (type *)(base + o); o = fig_align8(o + (long long)sizeof(type) * (count)); } while (0)
    FIG_PC(state, unsigned char, (long long)state_bytes);
    FIG_PC(best, unsigned char, capGg + 8);
    FIG_PC(orig, unsigned char, capGg + 8);
    FIG_PC(prev, unsigned char, capGg + 8);
    FIG_PC(saved, unsigned char, nU + 8);
    FIG_PC(org, int, 2LL * nU + 2);
    FIG_PC(ppos_org, int, 3LL * nP + 3);
    FIG_PC(repeatflag, int, 3LL * nP + 3);
    FIG_PC(used_read_arr, int, range + 2);
    FIG_PC(lrmd, int, 2LL * range + 4);
    FIG_PC(org_spec, int, 2LL * nU + 2);
    FIG_PC(ppos_spec, int, 3LL * nP + 3);
    long long stride = fig_align8((long long)sizeof(FigSlot)) + 2 * fig_align8(capGg + 8) + fig_align8(nU + 8);
    if (p) { p->slots = base + o; p->slot_stride = stride; }
    o += stride * nslots;
#undef FIG_PC
    return o;
}

FIG_D FigSlot *fig_slot_hdr(const FigPersist &P, int s) { return (FigSlot *)(P.slots + P.slot_stride * s); }
FIG_D unsigned char *fig_slot_cons(const FigPersist &P, int s) { return P.slots + P.slot_stride * s + fig_align8((long long)sizeof(FigSlot)); }
FIG_D unsigned char *fig_slot_prev(const FigPersist &P, int s, int capGg) { return fig_slot_cons(P, s) + fig_align8(capGg + 8); }
FIG_D unsigned char *fig_slot_mark(const FigPersist &P, int s, int capGg) { return fig_slot_cons(P, s) + 2 * fig_align8(capGg + 8); }

// LDS state <-> persistent snapshot (all lanes; ends with a barrier)
FIG_D void fig_state_save(FigEng &E, const FigPersist &P) {
    FIG_SYNC();
    const int n = (int)(sizeof(FigState) / 4);
    const int *src = (const int *)E.S; int *dst = (int *)P.state;
    for (int i = E.tid; i < n; i += E.nt) dst[i] = src[i];
    FIG_SYNC();
}
FIG_D void fig_state_load(FigEng &E, const FigPersist &P) {
    FIG_SYNC();
    const int n = (int)(sizeof(FigState) / 4);
    int *dst = (int *)E.S; const int *src = (const int *)P.state;
    for (int i = E.tid; i < n; i += E.nt) dst[i] = src[i];
    FIG_SYNC();
}

// Which arrays of FigScr point into the gap's persistent slab (FIG_BIND_SEQ: everything a gap carries;
// FIG_BIND_EVAL: speculative candidate -- results go to the slot, the G0 records to the *_spec buffers;
// FIG_BIND_REPLAY: bookkeeping reads the slot).
enum { FIG_BIND_SEQ = 0, FIG_BIND_EVAL = 1, FIG_BIND_REPLAY = 2 };
FIG_D void fig_bind(FigEng &E, const FigScr &work, const FigPersist &P, int mode, int slot, int capGg) {
    E.scr = work;
    E.scr.best = P.best; E.scr.orig = P.orig; E.scr.saved = P.saved;
    E.scr.repeatflag = P.repeatflag; E.scr.used_read_arr = P.used_read_arr; E.scr.lrmd = P.lrmd;
    E.scr.org = P.org; E.scr.ppos_org = P.ppos_org;
    if (mode == FIG_BIND_SEQ) { E.scr.prev = P.prev; }
    else {
        E.scr.cons = fig_slot_cons(P, slot); E.scr.mark = fig_slot_mark(P, slot, capGg);
        if (mode == FIG_BIND_EVAL) { E.scr.org = P.org_spec; E.scr.ppos_org = P.ppos_spec; }
    }
}

#define FIG_PSR_UNSET (-2)

// One speculative candidate: load the gap's snapshot, evaluate candidate j, write the slot.
template <bool LDS>
FIG_D void fig_spec_eval(FigEng &E, const FigScr &work, const FigPersist &P, int j, int slot, int capGg) {
    FIG_T0(E);
    fig_bind(E, work, P, FIG_BIND_EVAL, slot, capGg);
    fig_state_load(E, P);
    FIG_TICK(E, 34);
    FigState &S = *E.S;
    const unsigned long long flops0 = E.flops;
    if (E.tid == 0) {
        fig_slot_hdr(P, slot)->flops = 0; S.n_place = 0;
        S.L.j = j; S.L.gapEstimate = S.L.gapMin + j;
        S.psr_temp[0] = S.psr_temp[1] = FIG_PSR_UNSET;
        if (j != 0) S.prev_len = 0;             // a different length than this candidate's: never equal (see header)
    }
    FIG_SYNC();
    if (j == 0) for (int x = E.tid; x < S.prev_len; x += E.nt) E.scr.prev[x] = P.prev[x];
    FIG_SYNC();
    fig_eval_candidate<LDS>(E);
    FIG_TICK(E, 26);                                 // (everything inside fig_eval_candidate, which has its own slots)
    if (E.tid == 0) {
        FigSlot *h = fig_slot_hdr(P, slot);
        h->lik = S.L.likelihood; h->region_perct = S.region_perct; h->end_pos_max = S.end_pos_max;
        h->G = S.L.gapEstimate; h->iters = S.L.ev_iters; h->valid = S.valid_count; h->side_break = S.L.ev_side_break; h->fill = S.L.ev_fill;
        h->flags = (S.umaxleftf ? 1 : 0) | (S.umaxrightf ? 2 : 0) | (S.ucoverf ? 4 : 0) | (S.discont_or_not ? 8 : 0);
        h->psr0 = S.psr_temp[0]; h->psr1 = S.psr_temp[1];
        h->left = S.left; h->right = S.right; h->side_limit = S.side_limit; h->cons_len = S.cons_len;
        h->partial_read_len = S.partial_read_len; h->partial_read_count = S.partial_read_count; h->n_place = S.n_place; h->prev_len = S.prev_len;
    }
    FIG_SYNC();
    {   // previous_str as this candidate leaves it (:3919-3927): the next sequential step (the run(originalGap) fallbacks) compares against it
        unsigned char *pv = fig_slot_prev(P, slot, capGg);
        for (int x = E.tid; x < S.prev_len; x += E.nt) pv[x] = E.scr.prev[x];
    }
    {   // one atomic pair per wave, not per lane (a partial-mode candidate is ~1 ms of work: 512 same-address atomics showed up in its profile)
        unsigned long long d = E.flops - flops0;
#ifndef FIG_EMU
        for (int off = 32; off > 0; off >>= 1) d += fig_shfl_down_u64(d, off);
        if (E.lane != 0) d = 0;
#endif
        if (d) { fig_atomic_add_u64(&fig_slot_hdr(P, slot)->flops, d); fig_atomic_add_u64(&E.B->counters[2], d); }
    }
    E.flops = flops0;                             // speculative work is credited by the replay, and only if it is consumed
    FIG_SYNC();
    FIG_TICK(E, 39);
}

// Replay the bookkeeping of up to n speculated candidates (slots 0..n-1 hold candidates j0, j0+1, ...).
FIG_D void fig_spec_replay(FigEng &E, const FigScr &work, const FigPersist &P, int n, int capGg) {
    fig_bind(E, work, P, FIG_BIND_REPLAY, 0, capGg);
    fig_state_load(E, P);
    FigState &S = *E.S;
    const int G0 = E.g->G0, nU = E.g->nU, nP = E.g->nP;
    for (int s = 0; s < n && !S.L.done; s++) {
        fig_bind(E, work, P, FIG_BIND_REPLAY, s, capGg);
        const FigSlot *h = fig_slot_hdr(P, s);
        if (E.tid == 0) {
            S.L.likelihood = h->lik; S.region_perct = h->region_perct; S.end_pos_max = h->end_pos_max;
            S.G = h->G; S.ncols = h->G; S.L.ev_iters = h->iters; S.valid_count = h->valid; S.L.ev_side_break = h->side_break; S.L.ev_fill = h->fill;
            S.umaxleftf = h->flags & 1; S.umaxrightf = (h->flags >> 1) & 1; S.ucoverf = (h->flags >> 2) & 1; S.discont_or_not = (h->flags >> 3) & 1;
            if (h->psr0 != FIG_PSR_UNSET || h->psr1 != FIG_PSR_UNSET) { S.psr_temp[0] = h->psr0; S.psr_temp[1] = h->psr1; }
            S.left = h->left; S.right = h->right; S.side_limit = h->side_limit; S.cons_len = h->cons_len;
            S.partial_read_len = h->partial_read_len; S.partial_read_count = h->partial_read_count;
            S.n_place += h->n_place; S.flops_useful += h->flops;
            if (!h->side_break && !(h->fill != 0 && S.L.inr)) S.overlap_threshold = 5;      // this candidate's loop body got to Figbird.cpp:6317
        }
        FIG_SYNC();
        if (h->G == G0 && !h->side_break && !(h->fill != 0 && S.L.inr)) {     // this candidate's placeReads ran with gapLength == originalGap
            for (int i = E.tid; i < 2 * nU; i += E.nt) P.org[i] = P.org_spec[i];
            for (int i = E.tid; i < 3 * nP; i += E.nt) P.ppos_org[i] = P.ppos_spec[i];
        }
        if (!h->side_break && !(h->fill != 0 && S.L.inr) && S.num_itr > 0) {
            const unsigned char *pv = fig_slot_prev(P, s, capGg);
            for (int x = E.tid; x < h->prev_len; x += E.nt) P.prev[x] = pv[x];
            if (E.tid == 0) S.prev_len = h->prev_len;
        }
        fig_loop_step(E);
    }
    fig_state_save(E, P);
}
#endif
