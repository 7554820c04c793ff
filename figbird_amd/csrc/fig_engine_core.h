// fig_engine_core.h -- included by fig_engine.h.  placeReads, finalize and fillGap for one gap.
// Function names and the file:line citations follow the reference (Figbird.cpp).

// ---------------------------------------------------------------------------------------
// Placement window of a read: the offsets o = i - gapStart that pass the insert-size test
// (Figbird.cpp:3550-3557 unmapped, :3128-3135 / :3195-3203 partial).
struct FigWin { int lo, hi; int tis0; int dir; };       // tempInsertSize(o) = tis0 + dir*o

FIG_D FigWin fig_window_unmapped(const FigEng &E, int pos1, int len, int G, int gapoffset) {
    const FigDevModel &M = *E.M;
    long long gsl = E.g->gapStart;
    FigWin w;
    int lo = -(len - 1), hi = G - 1;
    if ((long long)pos1 < gsl) {
        w.tis0 = (int)(gsl - pos1 + len); w.dir = 1;     // tis = insertSize + i - gapStart
        int a = M.Tmin - w.tis0, b = M.Tmax - w.tis0;
        if (a > lo) lo = a;
        if (b < hi) hi = b;
    } else {
        int p = pos1 + gapoffset;
        w.tis0 = (int)(p - gsl + len); w.dir = -1;        // tis = insertSize + gapStart + G - i
        int a = w.tis0 - M.Tmax, b = w.tis0 - M.Tmin;
        if (a > lo) lo = a;
        if (b < hi) hi = b;
    }
    w.lo = lo; w.hi = hi;
    return w;
}

// partial read: left side o in [-(len-1), -1], right side o in [G-len+1, G-1]; the insert test only
// applies when the (possibly shifted) mate position is not -1.
FIG_D FigWin fig_window_partial(const FigEng &E, int pos1, int ref_pos, int len, int G, int gapoffset, int shift_always) {
    const FigDevModel &M = *E.M;
    long long gsl = E.g->gapStart;
    FigWin w;
    if ((long long)pos1 < gsl) {
        w.lo = -(len - 1); w.hi = -1;
        w.tis0 = (int)(gsl - ref_pos + len); w.dir = 1;
        if (ref_pos != -1) {
            int a = M.Tmin - w.tis0, b = M.Tmax - w.tis0;
            if (a > w.lo) w.lo = a;
            if (b < w.hi) w.hi = b;
        }
    } else {
        int rp = ref_pos;
        if (shift_always || rp != -1) rp += gapoffset;
        w.lo = G - len + 1; w.hi = G - 1;
        w.tis0 = (int)(rp - gsl + len); w.dir = -1;
        if (rp != -1) {
            int a = w.tis0 - M.Tmax, b = w.tis0 - M.Tmin;
            if (a > w.lo) w.lo = a;
            if (b < w.hi) w.hi = b;
        }
    }
    return w;
}

// E-step product of one placement (Figbird.cpp:3142-3167 / :3563-3589), bases j0..j1-1.  Generic
// (slow) form used by the partial-read path; the unmapped path uses fig_hot_estep().
FIG_D double fig_estep_chain(FigEng &E, double p, int o, int len, int j0, int j1, int rev, int G, int left, int right) {
    const double *e = E.M->e, *ome = E.M->ome1;
    (void)G; (void)right;
    for (int j = j0; j < j1; j++) {
        int x = o + j;
        if (x < -left) continue;
        int b = E.rb[j];
        int k = rev ? len - 1 - j : j;
        if (b < 4) { FigPQ v = fig_pq_get(E, b, x); p *= (v.p * ome[k] + e[k] * v.q); }
        else p *= (e[k] * fig_q4_get(E, x));
    }
    return p;
}

// MLE product of one placement (Figbird.cpp:3383-3407 / :3761-3785).
FIG_D double fig_mle_chain(FigEng &E, int o, int len, int j0, int j1, int rev, int G, int left, int right) {
    const double *e = E.M->e, *m3 = E.M->m3, *T = E.M->T;
    double q = 1;
    for (int j = j0; j < j1; j++) {
        int x = o + j;
        int to = E.rb[j];
        int from = fig_from_code(E, x, G, left, right);
        int k = rev ? len - j - 1 : j;
        if (from == to) q *= m3[k];
        else q *= e[k] * T[from * 5 + to];
    }
    return q;
}

// Add the weights of a chunk of partial reads to the gap columns in (read, placement) order (the
// `countsGap[i-startPos+j][...] += antilog_value` loops, Figbird.cpp:3181-3187).  Lanes = gap columns; rows t = 0..nr-1 of
// `wrows` hold the reads' weights by placement, `codes + t*256` their base codes, S.tm_* their windows.  A column's five
// sums are fetched from the slab once per chunk, every read adds into them in read order, and the 5-way choice is five
// select-adds (x + 0.0 == x exactly: the sums never go negative), so lanes with different bases do not serialise and the
// loads of consecutive placements overlap.
FIG_D void fig_accumulate_chunk(FigEng &E, int nr, const double *wrows, long long rstride, const unsigned char *codes, int G) {
    const FigState &S = *E.S;
    const int cg = E.capG;
    for (int x = E.tid; x < G; x += E.nt) {
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0;
        bool have = false;
        for (int t = 0; t < nr; t++) {
            const int lo = S.tm_lo[t], hi = S.tm_hi[t], len = S.tm_len[t];
            int o0 = x - (len - 1); if (o0 < lo) o0 = lo;
            int o1 = x; if (o1 > hi) o1 = hi;
            if (o0 > o1) continue;
            if (!have) { a0 = E.scr.cnt[x]; a1 = E.scr.cnt[cg + x]; a2 = E.scr.cnt[2 * cg + x]; a3 = E.scr.cnt[3 * cg + x]; a4 = E.scr.cnt[4 * cg + x]; have = true; }
            const double *wr = wrows + (long long)t * rstride - lo;
            const unsigned char *rb = codes + t * 256 + x;
#pragma unroll 4
            for (int o = o0; o <= o1; o++) {
                const double w = wr[o];
                const int b = rb[-o];
                a0 += b == 0 ? w : 0.0; a1 += b == 1 ? w : 0.0; a2 += b == 2 ? w : 0.0; a3 += b == 3 ? w : 0.0; a4 += b > 3 ? w : 0.0;
            }
        }
        if (have) { E.scr.cnt[x] = a0; E.scr.cnt[cg + x] = a1; E.scr.cnt[2 * cg + x] = a2; E.scr.cnt[3 * cg + x] = a3; E.scr.cnt[4 * cg + x] = a4; }
    }
}

// ---------------------------------------------------------------------------------------
// findRepeat, Figbird.cpp:1799-1911.  Lanes = partial reads.
FIG_D int fig_count_occ(const unsigned char *s, int slen, const unsigned char *pat, int plen, int *first, int *last) {
    int cnt = 0;
    for (int pos = 0; pos + plen <= slen; pos++) {
        int k = 0;
        while (k < plen && s[pos + k] == pat[k]) k++;
        if (k == plen) { if (cnt == 0) *first = pos; *last = pos; cnt++; }
    }
    return cnt;
}

FIG_D void fig_find_repeat(FigEng &E) {
    FigState &S = *E.S;
    int np = E.g->nP;                              // lines kept (<= partial_limit + 1)
    if (E.tid == 0) { S.ibuf[0] = 0; S.ibuf[1] = 0; }
    FIG_SYNC();
    const int n = 20;
    for (int p = E.tid; p < np; p += E.nt) {
        unsigned char rs[FIG_MAX_READLEN + 8];
        int len = fig_load_read(E, E.B->p, E.g->pBase + p, rs);
        int rf0 = -1, rf1 = -1, rf2 = -1, leftfound = 0, anyfound = 0, both = 0;
        int lim = S.gl_len - n;
        for (int i = 0; i < lim; i++) {
            int first = 0, last = 0;
            int c = fig_count_occ(rs, len, S.gap_left + i, S.gl_len - i, &first, &last);
            if (c > 1) { rf0 = 1; rf1 = S.gl_len - i; rf2 = first; leftfound = 1; anyfound = 1; break; }
        }
        lim = S.gr_len - n;
        for (int i = 0; i < lim; i++) {
            int first = 0, last = 0;
            int c = fig_count_occ(rs, len, S.gap_right, S.gr_len - i, &first, &last);
            if (c > 1) { rf0 = 2; rf1 = last; if (leftfound) both = 1; anyfound = 1; break; }
        }
        E.scr.repeatflag[p * 3] = rf0; E.scr.repeatflag[p * 3 + 1] = rf1; E.scr.repeatflag[p * 3 + 2] = rf2;
        if (anyfound) fig_atomic_or_i32(&S.ibuf[0], 1);
        if (both) fig_atomic_or_i32(&S.ibuf[1], 1);
    }
    FIG_SYNC();
    if (E.tid == 0) { S.one_side_repeat_flag = S.ibuf[0] ? 1 : S.one_side_repeat_flag; S.rep_flag = S.ibuf[1]; }
    FIG_SYNC();
}

// update_partial_prob with lanes = gap columns (no atomics on the counts, no per-base global loads): the reads' packed words
// and their column maps are staged once in the idle weight-row area of LDS (read p covers columns [ja, jb) with base index
// i = j + off: off = clip + 1 from the left flank, clip - gaplen from the right), then every column walks the reads.  The counts
// are integers, so the order is immaterial (as in the form below, which stays for gaps whose reads do not fit the area).
FIG_D bool fig_update_partial_prob_cols(FigEng &E, int gaplen) {
    FigState &S = *E.S;
    const int cg = E.capG, np = E.g->nP;
    const int RW = 28;                                             // dwords per staged read: 4 of column map + up to 13 + 7 packed words (+ pad)
    if (!E.w_lds || np < 1 || (long long)np * RW * 4 > (long long)E.nteams * E.Wcap * 8) return false;
    uint32_t *st = (uint32_t *)(fig_lds + E.off_w);
    const FigDevReads &PR = E.B->p;
    const long long pb = E.g->pBase;
    if (E.tid == 0) { S.left_max = -FIG_MAX_GAP; S.right_min = FIG_MAX_GAP; }
    FIG_SYNC();
    for (int p = E.tid; p < np; p += E.nt) {
        const long long idx = pb + p;
        const int len = PR.len[idx], match = PR.aux[idx];
        int clip = PR.clip[idx];
        fig_atomic_max_i32(&S.partial_read_len, len);
        const int rf = E.scr.repeatflag[p * 3];
        if (rf == 1) clip = E.scr.repeatflag[p * 3 + 2] + E.scr.repeatflag[p * 3 + 1] - 1;
        if (rf == 2) clip = E.scr.repeatflag[p * 3 + 1];
        int stop1 = len - clip - 1; if (stop1 > gaplen) stop1 = gaplen;
        const int stop2 = (clip <= gaplen) ? 0 : clip - gaplen;
        int ja = 0, jb = 0, off = 0;
        if (match == 1 || match == 4) {
            const int n = stop1 > 0 ? stop1 : 0;
            ja = 0; jb = n; off = clip + 1;
            fig_atomic_max_i32(&S.left_max, n - 1);
        } else if (match == 2 || match == 3) {
            const int n = clip - 1 >= stop2 ? clip - stop2 : 0;   // bases clip-1 .. stop2 -> columns gaplen-1 .. gaplen-n
            ja = gaplen - n; jb = gaplen; off = clip - gaplen;
            fig_atomic_min_i32(&S.right_min, gaplen - n);
        }
        uint32_t *r = st + (long long)p * RW;
        r[0] = (uint32_t)ja; r[1] = (uint32_t)jb; r[2] = (uint32_t)off; r[3] = (uint32_t)len;
    }
    for (int i = E.tid; i < np * 24; i += E.nt) {                  // packed 2-bit words then N-mask words of every read
        const int p = i / 24, k = i - p * 24;
        const int len = PR.len[pb + p];
        const int nw2 = (len + 15) >> 4, nwm = (len + 31) >> 5;
        st[(long long)p * RW + 4 + k] = k < nw2 + nwm ? E.B->packed[PR.woff[pb + p] + k] : 0u;
    }
    FIG_SYNC();
    const int left_max = S.left_max, right_min = S.right_min;
    const int rstart = right_min + 5 > left_max - 4 ? (right_min + 5 > 0 ? right_min + 5 : 0) : (left_max - 4 > 0 ? left_max - 4 : 0);   // first column of partial_right
    for (int j = E.tid; j < gaplen; j += E.nt) {
        int c0 = 1, c1 = 1, c2 = 1, c3 = 1;
        if (j < cg) {
            for (int p = 0; p < np; p++) {
                const uint32_t *r = st + (long long)p * RW;
                const int ja = (int)r[0], jb = (int)r[1];
                if (j < ja || j >= jb) continue;
                const int i = j + (int)r[2], len = (int)r[3];
                int c = 4;
                if (i >= 0 && i < len) {
                    const int nw2 = (len + 15) >> 4;
                    const uint32_t w = r[4 + (i >> 4)], m = r[4 + nw2 + (i >> 5)];
                    c = ((m >> (i & 31)) & 1) ? 4 : (int)((w >> ((i & 15) * 2)) & 3);
                }
                c0 += (c == 0 || c == 4); c1 += (c == 1 || c == 4); c2 += (c == 2 || c == 4); c3 += (c == 3 || c == 4);
            }
        }
        E.scr.pc[j] = c0; E.scr.pc[cg + j] = c1; E.scr.pc[2 * cg + j] = c2; E.scr.pc[3 * cg + j] = c3;
        const int total = c0 + c1 + c2 + c3;
        int max_index = 0, max_val = c0;
        if (c1 > max_val) { max_val = c1; max_index = 1; }
        if (c2 > max_val) { max_val = c2; max_index = 2; }
        if (c3 > max_val) { max_val = c3; max_index = 3; }
        E.scr.colchar[j] = (unsigned char)max_index;
        fig_pq_set_p(E, 0, j, (double)c0 / total); fig_pq_set_p(E, 1, j, (double)c1 / total); fig_pq_set_p(E, 2, j, (double)c2 / total); fig_pq_set_p(E, 3, j, (double)c3 / total);
        // partial_left / partial_right (:2062-2079: one pass over the columns, `i <= left_max - 5` -> left, else `i >= right_min + 5`
        // -> right, 99 characters each at most) are two contiguous column ranges: every column writes its own character
        if (j <= left_max - 5) { if (j < 99) S.partial_left[j] = (unsigned char)max_index; }
        else if (j >= right_min + 5) { const int k = j - rstart; if (k < 99) S.partial_right[k] = (unsigned char)max_index; }
    }
    if (E.tid == 0) {
        if (E.M->partial_flag == 1) S.partial_read_count = np;
        int nl = left_max - 4; if (nl < 0) nl = 0; if (nl > gaplen) nl = gaplen;
        int nr = gaplen - rstart; if (nr < 0) nr = 0;
        S.pl_len = nl < 99 ? nl : 99; S.pr_len = nr < 99 ? nr : 99;
    }
    FIG_SYNC();
    return true;
}

// update_partial_prob, Figbird.cpp:1913-2088.  Lanes = partial reads (integer counts, so the
// order of the increments is immaterial), then lanes = columns.
FIG_D void fig_update_partial_prob(FigEng &E, int gaplen) {
    if (fig_update_partial_prob_cols(E, gaplen)) return;
    FigState &S = *E.S;
    int cg = E.capG, np = E.g->nP;
    for (int i = E.tid; i < gaplen; i += E.nt) { E.scr.pc[i] = 1; E.scr.pc[cg + i] = 1; E.scr.pc[2 * cg + i] = 1; E.scr.pc[3 * cg + i] = 1; }
    if (E.tid == 0) { S.left_max = -FIG_MAX_GAP; S.right_min = FIG_MAX_GAP; }
    FIG_SYNC();
    for (int p = E.tid; p < np; p += E.nt) {
        long long idx = E.g->pBase + p;
        int len = E.B->p.len[idx], clip = E.B->p.clip[idx], match = E.B->p.aux[idx];
        long long woff = E.B->p.woff[idx];
        fig_atomic_max_i32(&S.partial_read_len, len);
        int rf = E.scr.repeatflag[p * 3];
        if (rf == 1) clip = E.scr.repeatflag[p * 3 + 2] + E.scr.repeatflag[p * 3 + 1] - 1;
        if (rf == 2) clip = E.scr.repeatflag[p * 3 + 1];
        int stop1 = len - clip - 1; if (stop1 > gaplen) stop1 = gaplen;
        int stop2 = (clip <= gaplen) ? 0 : clip - gaplen;
        if (match == 1 || match == 4) {
            int j = 0;
            for (int i = clip + 1; i < clip + 1 + stop1; i++, j++) {
                int c = (i >= 0 && i < len) ? fig_read_code(E.B->packed, woff, len, i) : 4;
                if (c < 4) fig_atomic_add_i32(&E.scr.pc[c * cg + j], 1);
                else for (int h = 0; h < 4; h++) fig_atomic_add_i32(&E.scr.pc[h * cg + j], 1);
            }
            fig_atomic_max_i32(&S.left_max, j - 1);
        } else if (match == 2 || match == 3) {
            int j = gaplen - 1;
            for (int i = clip - 1; i >= stop2; i--, j--) {
                int c = (i >= 0 && i < len) ? fig_read_code(E.B->packed, woff, len, i) : 4;
                if (j < 0 || j >= cg) continue;
                if (c < 4) fig_atomic_add_i32(&E.scr.pc[c * cg + j], 1);
                else for (int h = 0; h < 4; h++) fig_atomic_add_i32(&E.scr.pc[h * cg + j], 1);
            }
            fig_atomic_min_i32(&S.right_min, j + 1);
        }
    }
    FIG_SYNC();
    if (E.tid == 0 && E.M->partial_flag == 1) S.partial_read_count = np;
    for (int i = E.tid; i < gaplen; i += E.nt) {
        int total = 0, max_index = 0, max_val = -1;
        for (int k = 0; k < 4; k++) { int v = E.scr.pc[k * cg + i]; total += v; if (v > max_val) { max_val = v; max_index = k; } }
        E.scr.colchar[i] = (unsigned char)max_index;
        for (int k = 0; k < 4; k++) fig_pq_set_p(E, k, i, (double)E.scr.pc[k * cg + i] / total);
    }
    FIG_SYNC();
    if (E.tid == 0) {
        int lc = 0, rc = 0;
        for (int i = 0; i < gaplen; i++) {
            if (i <= S.left_max - 5) { if (lc < 99) S.partial_left[lc++] = E.scr.colchar[i]; }
            else if (i >= S.right_min + 5) { if (rc < 99) S.partial_right[rc++] = E.scr.colchar[i]; }
        }
        S.pl_len = lc; S.pr_len = rc;
    }
    FIG_SYNC();
}

// ---------------------------------------------------------------------------------------
// find_contig_match (negative-overlap check, A9), Figbird.cpp:2176-2267.  Lane 0.
FIG_D int fig_find_contig_match(FigEng &E) {
    FigState &S = *E.S;
    if (E.g->G0 > E.M->neg_overlap) return 0;
    const int n = 3;
    int s1n = S.gl_len, s2n = S.gr_len;
    int np = E.g->nP;
    for (int i = 0; i < S.side_limit - n; i++) {
        if (i > s1n || i > s2n) break;
        int s3n = s1n - i, s4n = s2n - i;
        // pos = s3.find(s4)
        int found = -1;
        for (int pos = 0; pos + s4n <= s3n; pos++) {
            int k = 0;
            while (k < s4n && S.gap_left[i + pos + k] == S.gap_right[k]) k++;
            if (k == s4n) { found = pos; break; }
        }
        if (found != -1) {
            int remn = s2n - s4n;                     // rem_str = s2.substr(s4.size())
            int part_count = 0;
            for (int p = 0; p < np; p++) {
                unsigned char rs[FIG_MAX_READLEN + 8];
                int len = fig_load_read(E, E.B->p, E.g->pBase + p, rs);
                int max_match = -1, max_pos = -1;
                if (len >= s1n) {
                    for (int j = 0; j < len - s1n; j++) {
                        int mc = 0, mm = 0;
                        for (int kk = 0; kk < s1n; kk++) { if (rs[j + kk] == S.gap_left[kk]) mc++; else mm++; if (mm > 2) break; }
                        if (mc > max_match) { max_match = mc; max_pos = j; }
                    }
                }
                if (s1n - max_match <= 2) {
                    int newpos = max_pos + s1n, match = 0;
                    for (int j = 0; j < remn; j++) {
                        int rc = (newpos + j < len) ? rs[newpos + j] : 255;
                        if (S.gap_right[s4n + j] == rc) match++;
                    }
                    if (remn - match <= 2) return s4n;
                }
                part_count++;
                if (part_count > FIG_READ_CAP) break;
            }
        }
    }
    return 0;
}

// initialize, Figbird.cpp:2298-2383.  Returns (broadcast) the negative-overlap length.
FIG_D int fig_initialize(FigEng &E, int gl, int negGapCheck) {
    FigState &S = *E.S;
    int cg = E.capG;
    if (E.tid == 0) { S.G = gl; fig_ise(E); S.region_perct = 0; }
    FIG_SYNC();
    int G = S.G;
    for (int x = E.tid; x < cg; x += E.nt) {
        if (x < G) { E.scr.cnt[x] = 0; E.scr.cnt[cg + x] = 0; E.scr.cnt[2 * cg + x] = 0; E.scr.cnt[3 * cg + x] = 0; E.scr.cnt[4 * cg + x] = 1;
                     for (int j = 0; j < 5; j++) E.scr.ncnt[j * cg + x] = 0; }
        for (int j = 0; j < 5; j++) E.scr.tmp[j * cg + x] = 0;      // qual_gap
        E.scr.cov[x] = 0;
    }
    int nml = E.g->nU > E.g->nP ? E.g->nU : E.g->nP;
    for (int r = E.tid; r < nml; r += E.nt) E.scr.maxlv[r] = 0;
    for (int r = E.tid; r < E.g->nU; r += E.nt) { E.scr.hint[r] = FIG_NOPOS; E.scr.hint_e[r] = FIG_NOPOS; }
    FIG_SYNC();
    if (E.tid == 0) {
        int gfp = 0;
        if (S.side_limit > 0 && negGapCheck == 0) gfp = fig_find_contig_match(E);
        S.bc_i = gfp;
    }
    FIG_SYNC();
    int gfp = S.bc_i;
    // computeProbsGap(1): gap columns carry only N counts -> probs .25, then the partial-read seeding
    fig_compute_probs(E);
    FIG_SYNC();
    fig_update_partial_prob(E, G);
    fig_compute_errprobs_only(E);
    FIG_SYNC();
    return gfp;
}

// ---------------------------------------------------------------------------------------
// helpers on code strings (lane 0)
FIG_D double fig_get_diff(const unsigned char *target, const unsigned char *ref, int length) {   // :2429-2443
    double diff = 0;
    for (int i = 0; i < length; i++) if (target[i] != ref[i]) diff++;
    return diff / length;
}
// find_partial_match, :2445-2490
FIG_D int fig_find_partial_match(const unsigned char *ref, int len_r, const unsigned char *search, int len_s, int pos, int c, int len_t) {
    int len_thresh = (c == 0) ? len_t : 4;
    if (len_r >= len_s && len_s >= len_thresh) {
        double frac = (pos == 0) ? fig_get_diff(ref + (len_r - len_s), search, len_s) : fig_get_diff(ref, search, len_s);
        if (frac < 0.2 && c == 1) return 1;
        if (frac <= 0.08 && c == 0) return 1;
    }
    return 0;
}
// substr with std::string clamping; pos > size yields an empty view (the reference would throw).
FIG_D int fig_substr(const unsigned char *s, int slen, int pos, int n, const unsigned char **out) {
    if (pos < 0 || pos > slen) { *out = s; return 0; }
    int m = slen - pos;
    if (n < 0 || n > m) n = m;
    *out = s + pos;
    return n;
}
// get_read_frag, :2492-2511
FIG_D int fig_get_read_frag(const FigEng &E, const unsigned char *s1, int len, int placed_pos, const unsigned char **out) {
    int sl = E.S->side_limit, neg = -placed_pos;
    if (placed_pos < 0) {
        if (neg < sl) return fig_substr(s1, len, 0, neg, out);
        return fig_substr(s1, len, neg - sl, sl, out);
    }
    return fig_substr(s1, len, E.S->G - placed_pos, sl, out);
}

// detect_overlap_gapestimate (A10), Figbird.cpp:2513-2779.  Lane 0.  pflag rows are
// {used, placed position, ...} with the given stride.
FIG_D void fig_detect_overlap(FigEng &E, const int *pflag, int stride, int gaplen, int *ret_v, int len_thresh) {
    FigState &S = *E.S;
    int l_max = -FIG_MAX_GAP, r_min = FIG_MAX_GAP;
    int left_c = 0, right_c = 0, overlap_count = 0;
    const double mismatch_threshold = .1;
    int prc = S.partial_read_count, np = E.g->nP;
    const FigDevReads &PR = E.B->p;
    long long pb = E.g->pBase;
    int pc = 0;
    for (int k = 0; k < np; k++) {
        if (pc >= prc) break;
        int match = PR.aux[pb + k], len = PR.len[pb + k];
        E.scr.smflag[pc] = 0;
        if (pflag[pc * stride] == 0) { pc++; if (pc > FIG_READ_CAP) break; continue; }
        int pos = pflag[pc * stride + 1];
        int j, flag = 0, start = -1;
        for (j = 0; j < len; j++) {
            if (pos + j >= 0 && pos + j < gaplen) { if (flag == 0) { flag = 1; start = pos + j; } }
            if (pos + j == gaplen) break;
        }
        if (match == 1 || match == 4 || pos < 0) { if (pos + j - 1 > l_max) l_max = pos + j - 1; }
        else { if (start < r_min) r_min = start; }
        pc++;
        if (pc > FIG_READ_CAP) break;
    }
    if (l_max == -FIG_MAX_GAP) l_max = -1;
    if (r_min == FIG_MAX_GAP) r_min = gaplen;
    overlap_count = l_max - r_min + 1;
    int ovflag = 0;
    int G = S.G, sl = S.side_limit;
    unsigned char rs1[FIG_MAX_READLEN + 8], rs2[FIG_MAX_READLEN + 8];
    for (int k = 0; k < prc; k++) {
        E.scr.smflag[k] = 0;
    }
    for (int k = 0; k < prc; k++) {
        if (pflag[k * stride] == 1) {
            int len = PR.len[pb + k], mr = PR.aux[pb + k];
            int placed_pos = pflag[k * stride + 1];
            if (placed_pos < 0 && placed_pos + len > G) {
                fig_load_read(E, PR, pb + k, rs1);
                const unsigned char *l, *r;
                int ln = (-placed_pos < sl) ? fig_substr(rs1, len, 0, -placed_pos, &l) : fig_substr(rs1, len, -placed_pos - sl, sl, &l);
                int rn = fig_substr(rs1, len, -placed_pos + G, sl, &r);
                if (fig_find_partial_match(S.gap_left, S.gl_len, l, ln, 0, 0, len_thresh) &&
                    fig_find_partial_match(S.gap_right, S.gr_len, r, rn, 1, 0, len_thresh)) ovflag = 1;
            }
            if (placed_pos < 0 && placed_pos + len - 1 >= r_min && placed_pos + len <= G) E.scr.lcross[left_c++] = k;
            if (placed_pos > 0 && placed_pos <= l_max) E.scr.rcross[right_c++] = k;
            if (placed_pos < 0 && placed_pos + len > G && (mr == 2 || mr == 3)) { E.scr.rcross[right_c++] = k; E.scr.smflag[k] = 1; }
        }
    }
    if (ovflag || (E.g->stat2 == 1 && E.g->G0 <= 20 && gaplen == E.g->stat3)) { ret_v[0] = 300; ret_v[1] = 0; return; }
    if (r_min <= l_max) {
        int max_overlap = 0, false_overlap_flag = 0;
        for (int i = 0; i < left_c; i++) {
            int li = E.scr.lcross[i];
            int len = fig_load_read(E, PR, pb + li, rs1);
            int p1 = pflag[li * stride + 1];
            for (int j = 0; j < right_c; j++) {
                int rj = E.scr.rcross[j];
                if (li == rj) continue;
                int p2 = pflag[rj * stride + 1];
                int diff_gap = p1 + len - gaplen;
                if (diff_gap > 0) overlap_count = (p1 + len - 1) - p2 + 1 - diff_gap;
                else { overlap_count = (p1 + len - 1) - p2 + 1; diff_gap = 0; }
                if (overlap_count >= S.overlap_threshold) {
                    int len2 = fig_load_read(E, PR, pb + rj, rs2);
                    const unsigned char *cl = rs1, *cr = rs2, *fr;
                    int len1 = 0, lenr = 0, fn;
                    if (E.scr.smflag[rj] != 1) {
                        fn = fig_get_read_frag(E, rs1, len, p1, &fr);
                        if (fig_find_partial_match(S.gap_left, S.gl_len, fr, fn, 0, 1, -1))
                            len1 = fig_substr(rs1, len, len - overlap_count - diff_gap, overlap_count, &cl);
                        fn = fig_get_read_frag(E, rs2, len2, p2, &fr);
                        if (fig_find_partial_match(S.gap_right, S.gr_len, fr, fn, 1, 1, -1))
                            lenr = fig_substr(rs2, len2, 0, overlap_count, &cr);
                    } else {
                        int x = p2;
                        fn = fig_get_read_frag(E, rs1, len, p1, &fr);
                        if (fig_find_partial_match(S.gap_left, S.gl_len, fr, fn, 0, 1, -1))
                            len1 = fig_substr(rs1, len, len - overlap_count - x, overlap_count - x, &cl);
                        fn = fig_substr(rs2, len2, -x + G, sl, &fr);
                        if (fig_find_partial_match(S.gap_right, S.gr_len, fr, fn, 1, 1, -1))
                            lenr = fig_substr(rs2, len2, -x, overlap_count + x, &cr);
                    }
                    if (len1 > 0 && lenr > 0 && len1 == lenr) {
                        double mismatch_frac = fig_get_diff(cl, cr, len1);
                        if (mismatch_frac <= mismatch_threshold) {
                            if (len1 > max_overlap) { max_overlap = len1; S.psr_temp[0] = li; S.psr_temp[1] = rj; }
                        } else false_overlap_flag = -1;
                    }
                }
            }
        }
        if ((false_overlap_flag == 0 && max_overlap >= S.overlap_threshold) ||
            (false_overlap_flag == -1 && max_overlap >= 2 * S.overlap_threshold)) { ret_v[0] = max_overlap; ret_v[1] = 0; }
        else if (false_overlap_flag == -1 || max_overlap < S.overlap_threshold) { ret_v[0] = 0; ret_v[1] = -1; S.psr_temp[0] = S.psr_temp[1] = -1; }
        return;
    }
    ret_v[0] = 0; ret_v[1] = 0;
}

// findOverlapUnmapped, Figbird.cpp:2945-3019.  Lane 0.
// Where the lane-0 sorts of the post-processing keep their (pos, len, index) triples: the weight rows in LDS when they hold them
// (the rows are idle outside the E-step; a one-lane introsort makes ~n log n dependent accesses, and an LDS round trip is a
// fifth of one to the scratch slab), else the slab.
FIG_D FigTrip *fig_sort_buf(FigEng &E, int n) {
    if (E.w_lds && (long long)n * (long long)sizeof(FigTrip) <= (long long)E.nteams * (long long)E.Wcap * 8LL) return (FigTrip *)E.wbuf;
    return E.scr.sortbuf;
}

FIG_D double fig_find_overlap_unmapped(FigEng &E) {
    FigState &S = *E.S;
    int n = E.g->nU;
    FigTrip *vec = fig_sort_buf(E, n);
    for (int i = 0; i < n; i++) {
        vec[i].v[0] = E.scr.frp[i * 2]; vec[i].v[1] = E.scr.frp[i * 2 + 1];
        vec[i].v[2] = i;
    }
    fig_std_sort(vec, n);
    int incorrect_penalty = 0, gap_penalty = 0;
    for (int i = 0; i < n - 1; i++) {
        if (vec[i].v[0] != -200) {
            int diff = (vec[i].v[0] + vec[i].v[1] - vec[i + 1].v[0]);
            if (diff >= 4) {
            } else if (diff >= 0) { incorrect_penalty += -250; S.discont_or_not = 1; }
            else {
                gap_penalty += -4 * 50;
                int a = vec[i].v[2], b = vec[i + 1].v[2];
                E.scr.mark[a] = 0; E.scr.mark[b] = 0;
                if (S.G == E.g->G0) { E.scr.org[a * 2] = -200; E.scr.org[b * 2] = -200; E.scr.org[a * 2 + 1] = 0; E.scr.org[b * 2 + 1] = 0; }
                S.valid_count -= 2;
            }
        }
    }
    int lr = 0;
    for (int i = 0; i < n; i++) {
        if (vec[i].v[0] != -200) {
            if (vec[i].v[0] < 0 && -vec[i].v[0] >= 3 && vec[i].v[0] + vec[i].v[1] > 0) lr++;
            if (vec[i].v[0] < S.G && vec[i].v[0] + vec[i].v[1] - S.G >= 3) lr++;
        }
    }
    return (double)(incorrect_penalty + gap_penalty + lr * 50);
}

// seed re-weighting of countsGap from the edges of the N region, Figbird.cpp:4032-4376.  Lane 0.
FIG_D void fig_seed_reweight(FigEng &E) {
    FigState &S = *E.S;
    int cg = E.capG, G = S.G, nU = E.g->nU;
    int url = E.M->read_length;
    int index_pair[2] = {-1, -1};
    int start_n = 0, pair_count = 0, N_count = 0, num_match_count0 = 0;
    int match_threshold = (int)(url * 0.25), max_segment_length = (int)(url * 0.67), min_gap_len = url / 2 + 1;
    double *count_pos = E.scr.tmp;                   // [4][cg], column-major by base
    for (int j = 0; j < G; j++) for (int k = 0; k < 4; k++) count_pos[k * cg + j] = 0;
    const unsigned char *con = E.scr.cons;
    int clen = S.cons_len;
    // The reference pushes (start, end) pairs on index_pair[1000] and pops a pair straight after its
    // end is pushed when the N run is shorter than min_gap_len (:4059-4088); only slots [0], [1] and the
    // top survivor are read afterwards (:4091-4107).  Track exactly those.
    int last_val = -1;
    for (int i = 0; i < clen; i++) {
        if (con[i] == 4 && start_n == 0) {
            start_n = 1;
            int v = i > 0 ? i - 1 : i;
            if (pair_count < 2) index_pair[pair_count] = v;
            pair_count++;
            N_count++;
        } else if (con[i] != 4 && start_n == 1) {
            start_n = 0;
            if (pair_count < 2) index_pair[pair_count] = i;
            pair_count++;
            if (N_count < min_gap_len) pair_count -= 2; else last_val = i;
            N_count = 0;
        } else if (con[i] == 4 && start_n == 1) N_count++;
        if (i == clen - 1 && start_n == 1) {
            if (pair_count < 2) index_pair[pair_count] = i;
            pair_count++;
            if (N_count < min_gap_len) pair_count -= 2; else last_val = i;
        }
    }
    if (pair_count > 2) index_pair[1] = last_val;
    int flag1 = 1, flag2 = 1;
    if (pair_count < 2) { flag1 = 0; flag2 = 0; }
    if (!(flag1 == 1 || flag2 == 1)) return;
    int end_index1 = index_pair[0], start_index1 = index_pair[1];
    int index_s = (end_index1 >= max_segment_length) ? end_index1 - max_segment_length + 1 : 0;
    unsigned char text_left[FIG_MAX_READLEN + 8], text_right[FIG_MAX_READLEN + 8];
    int tln = 0, trn = 0;
    for (int j = index_s; j < clen && con[j] != 4 && tln < FIG_MAX_READLEN; j++) text_left[tln++] = con[j];
    int stop_index = (start_index1 + max_segment_length <= clen) ? start_index1 + max_segment_length - 1 : clen - 1;
    for (int j = start_index1; j <= stop_index && trn < FIG_MAX_READLEN; j++) text_right[trn++] = con[j];
    int temp_flag1 = flag1, temp_flag2 = flag2;
    unsigned char rd[FIG_MAX_READLEN + 8];
    for (int r = 0; r < nU; r++) {
        if (E.scr.mark[r] == 0 && E.scr.maxlv[r] != 0) {
            int R2 = fig_load_read(E, E.B->u, E.g->uBase + r, rd);
            int mapped_pos = E.B->u.pos[E.g->uBase + r];
            flag1 = temp_flag1; flag2 = temp_flag2;
            for (int gg = 0; gg < 2; gg++) {
                int placed_pos = (int)(E.g->gapStart + index_pair[gg]);
                int insertsize;
                if ((long long)mapped_pos < E.g->gapStart) insertsize = placed_pos + R2 - mapped_pos;
                else insertsize = mapped_pos - placed_pos + R2;
                if (insertsize < E.M->Tmin + 100 || insertsize > E.M->Tmax - 100) { if (gg == 0) flag1 = 0; else flag2 = 0; }
            }
            if (flag1 == 1) {
                for (int j = 0; j < tln; j++) {
                    int match = 0, match_flag = 1;
                    for (int k = 0; k < tln - j; k++) {
                        int rc = k < R2 ? rd[k] : 255;
                        if (text_left[j + k] != rc) { match_flag = 0; break; } else match++;
                    }
                    if (match_flag == 1 && match > match_threshold) {
                        num_match_count0++;
                        for (int q = 0; q < R2; q++) {
                            int ind_read = rd[q];
                            int ind_ref = index_pair[0] - match + 1 + q;
                            if (ind_ref == G) break;
                            if (ind_ref > index_pair[0]) {
                                if (ind_read < 4) count_pos[ind_read * cg + ind_ref] += match;
                                else for (int z = 0; z < 4; z++) count_pos[z * cg + ind_ref] += match;
                            }
                        }
                        break;
                    }
                }
            }
            if (flag2 == 1) {
                for (int j = 0; j < trn; j++) {
                    int match = 0, match_flag = 1;
                    for (int k = 0; k < trn - j; k++) {
                        int rc = k < R2 ? rd[R2 - 1 - k] : 255;
                        if (text_right[trn - 1 - (j + k)] != rc) { match_flag = 0; break; } else match++;
                    }
                    if (match_flag == 1 && match > match_threshold) {
                        for (int q = 0; q < R2; q++) {
                            int ind_read = rd[R2 - q - 1];
                            int ind_ref = index_pair[1] + match - 1 - q;
                            if (ind_ref < 0) break;
                            if (ind_ref < index_pair[1]) {
                                if (ind_ref >= G) continue;
                                if (ind_read < 4) count_pos[ind_read * cg + ind_ref] += match;
                                else for (int z = 0; z < 4; z++) count_pos[z * cg + ind_ref] += match;
                            }
                        }
                        break;
                    }
                }
            }
        }
    }
    if (flag1 == 1 && num_match_count0 == 0 && index_pair[0] < S.pl_len) {
        for (int f = index_pair[0] + 1; f < S.pl_len; f++) if (f < G) count_pos[S.partial_left[f] * cg + f] += 1;
    }
    // the right-hand twin (:4321-4331) is unreachable: num_match_count1 starts at 1 (:4040)
    for (int j = 0; j < G; j++) {
        int total_count_p = 0;
        for (int k = 0; k < 4; k++) total_count_p += count_pos[k * cg + j];     // int += double, truncating each step
        if (total_count_p > 0) for (int k = 0; k < 4; k++) E.scr.cnt[k * cg + j] = (count_pos[k * cg + j] / total_count_p);
    }
}

#include "fig_engine_hot.h"
#include "fig_engine_partial.h"

// Numeric planes (i)/(ii) of the parity contract for the candidate being evaluated (parity tests only; all lanes).
FIG_D void fig_dbg_planes(FigEng &E, int nreads) {
    const FigDevBatch &B = *E.B;
    if (!B.dbg_counts && !B.dbg_read_maxlv) return;
    FIG_SYNC();
    const int j = E.S->dbg_j;
    if (j >= 0 && j < B.dbg_max_cand) {
        const long long rec = (long long)E.g->gapNo * B.dbg_max_cand + j;
        if (B.dbg_counts) {
            int n = E.S->ncols < B.dbg_plane_cols ? E.S->ncols : B.dbg_plane_cols;
            double *dst = B.dbg_counts + rec * B.dbg_plane_cols * 5;
            for (int x = E.tid; x < n; x += E.nt) for (int b = 0; b < 5; b++) dst[(long long)x * 5 + b] = E.scr.cnt[b * E.capG + x];
        }
        if (B.dbg_read_maxlv) {
            int n = nreads < B.dbg_plane_reads ? nreads : B.dbg_plane_reads;
            double *dst = B.dbg_read_maxlv + rec * B.dbg_plane_reads;
            for (int r = E.tid; r < n; r += E.nt) dst[r] = E.scr.maxlv[r];
        }
    }
    FIG_SYNC();
}

// ---------------------------------------------------------------------------------------
// placeReads, Figbird.cpp:3022-4387.  Returns maxLikelihood in S.lik (after a barrier).
template <bool LDS>
FIG_D void fig_place_reads(FigEng &E, int ge, int finalize_flag, int gapoffset, int updateflag) {
    FigState &S = *E.S;
    const FigDevModel &M = *E.M;
    int cg = E.capG, G = S.G, left = S.left, right = S.right;
    int nU = E.g->nU, G0 = E.g->G0;
    if (E.tid == 0) S.n_place++;
    FIG_T0(E);
    for (int x = E.tid; x < S.ncols; x += E.nt) for (int j = 0; j < 5; j++) E.scr.cnt[j * cg + x] = 0;
    for (int r = E.tid; r < nU; r += E.nt) {
        E.scr.maxlv[r] = 0; E.scr.mark[r] = 0; E.scr.frp[r * 2] = -200; E.scr.frp[r * 2 + 1] = 0;
        if (G == G0) { E.scr.org[r * 2] = -200; E.scr.org[r * 2 + 1] = 0; }
    }
    for (int r = E.tid; r < S.partial_read_count; r += E.nt) E.scr.maxlv[r] = 0;
    if (E.tid == 0) S.lik = 0;
    FIG_SYNC();
    double maxLikelihood = 0;                        // meaningful on lane 0 only

    if (M.partial_flag) {
        int np = E.g->nP;
        int nproc = np < FIG_READ_CAP ? np : FIG_READ_CAP;           // reads 0..2999 (:3121-3122)
        const FigDevReads &PR = E.B->p;
        long long pb = E.g->pBase;
        // E-step over the partial reads (:3082-3264): one read per wave (a read has at most len-1 placements), up to `nrow`
        // reads per chunk; each wave stages its read's codes and its weight row, then all lanes add the chunk's rows
        // into the gap columns read by read (the reference's order), and lane 0 adds the per-read maxima in read order.
        unsigned char *const rb_keep = E.rb; double *const wbuf_keep = E.wbuf;
        unsigned long long pfl = 0, pml = 0;           // lane-local flop count of the partial passes (pml: the MLE share)
        // a partial read has at most len-1 <= 199 placements, so its weight row needs 208 doubles, not a full Wcap row:
        // every wave gets a row whenever the weight area (nteams x Wcap) holds nw such rows
        int nrow = E.nw < FIG_PLB_TEAMS ? E.nw : FIG_PLB_TEAMS;
        long long rstride = 208;
        if ((long long)nrow * rstride > (long long)E.nteams * E.Wcap) { nrow = E.nw < E.nteams ? E.nw : E.nteams; if (nrow > FIG_PLB_TEAMS) nrow = FIG_PLB_TEAMS; rstride = E.Wcap; }
        if (nrow < 1) nrow = 1;
        FIG_TICK(E, 31);
        // the form of fig_engine_partial.h where it applies (LDS table, no N base in the gap's partial reads, no clipping by the
        // contig start); otherwise the generic form below
        const bool pfast = fig_partial_fast<LDS>(E);
        if constexpr (LDS) { if (pfast) fig_partial_estep<LDS>(E, gapoffset, nproc, maxLikelihood, pfl); }
        for (int p0 = 0; p0 < nproc && !pfast; p0 += nrow) {
            const int nr = nproc - p0 < nrow ? nproc - p0 : nrow;
            if (E.wave < nr) {
                const int p = p0 + E.wave;
                unsigned char *rbw = (unsigned char *)E.plb + E.wave * 256;
                const int len = PR.len[pb + p], flag1 = PR.aux[pb + p];
                { const long long woff = PR.woff[pb + p]; for (int j = E.lane; j < len; j += E.wsz) rbw[j] = (unsigned char)fig_read_code(E.B->packed, woff, len, j); }
                E.rb = rbw;
                double *wr = wbuf_keep + (long long)E.wave * rstride;
                int j0 = (flag1 == 1 || flag1 == 4) ? 2 : 0, j1 = len - ((flag1 == 1 || flag1 == 4) ? 0 : 2);
                FigWin w = fig_window_partial(E, PR.pos[pb + p], PR.refpos[pb + p], len, G, gapoffset, 0);
                FigBest best; best.v = -FIG_DBL_MAX; best.o = FIG_NOPOS;
                // reads without N whose placements stay inside the flank window take the table/scalar-load form of the chain
                bool slow = left < E.xoff || (LDS && E.tiles > 0);      // LDS-tiled class: the LDS image holds one tile, take the chain on E.pq
                { bool n = false; for (int j = E.lane; j < len; j += E.wsz) n = n || rbw[j] > 3; slow = slow || fig_wave_any(n); }
                const FigPQ *PQt = fig_pq_ptr<LDS>(E);
                fig_cu32p pkr = (fig_cu32p)(fig_uptr(E.B->packed) + fig_u64(PR.woff[pb + p]));
                fig_cdp ktf = (fig_cdp)fig_uptr(E.kt_fwd);
                const int ncolE_u = fig_u(E.ncolE), xoff_u = fig_u(E.xoff), j0u = fig_u(j0), j1u = fig_u(j1);
                for (int o = w.lo + E.lane; o <= w.hi; o += E.wsz) {
                    double t;
                    if (slow) t = fig_estep_chain(E, 1.0, o, len, j0, j1, 0, G, left, right);
                    else {
                        t = 1.0;
                        const FigPQ *col = PQt + o + xoff_u;
                        int j = j0u;
                        for (; j + 4 <= j1u; j += 4) {
                            FigPQ v[4]; double kk[8];
#pragma unroll
                            for (int q = 0; q < 4; q++) { const int jj = j + q; const int b = (int)((pkr[jj >> 4] >> ((jj & 15) * 2)) & 3); v[q] = col[b * ncolE_u + jj]; kk[2 * q] = ktf[2 * jj]; kk[2 * q + 1] = ktf[2 * jj + 1]; }
#pragma unroll
                            for (int q = 0; q < 4; q++) t *= (v[q].p * kk[2 * q] + kk[2 * q + 1] * v[q].q);
                        }
                        for (; j < j1u; j++) { const int b = (int)((pkr[j >> 4] >> ((j & 15) * 2)) & 3); const FigPQ v = col[b * ncolE_u + j]; t *= (v.p * ktf[2 * j] + ktf[2 * j + 1] * v.q); }
                    }
                    t = fig_log(t);
                    if (t > best.v) { best.v = t; best.o = o; }
                    wr[o - w.lo] = fig_pow10(t);
                    pfl += 4ULL * (unsigned long long)(j1 - j0) + (unsigned long long)fig_ovl(o, len, G);
                }
                best = fig_wave_best(E, best);
                if (E.lane == 0) { S.wv_v[E.wave] = best.v; S.wv_o[E.wave] = best.o; S.tm_lo[E.wave] = w.lo; S.tm_hi[E.wave] = w.hi; S.tm_len[E.wave] = len; }
            }
            FIG_TICK(E, 27);
            FIG_SYNC();
            FIG_TICK(E, 28);
            fig_accumulate_chunk(E, nr, wbuf_keep, rstride, (const unsigned char *)E.plb, G);
            for (int t = 0; t < nr; t++) {
                if (E.tid == 0 && S.wv_o[t] != FIG_NOPOS) maxLikelihood += S.wv_v[t];
                if (E.tid == 0 && E.B->dbg_read_maxlv) E.scr.maxlv[p0 + t] = S.wv_o[t] != FIG_NOPOS ? S.wv_v[t] : 0.0;   // parity plane (ii); unused otherwise
            }
            FIG_TICK(E, 29);
            FIG_SYNC();
            FIG_TICK(E, 30);
        }
        E.rb = rb_keep; E.wbuf = wbuf_keep;
        FIG_TICK(E, 4);
        fig_dbg_planes(E, nproc);
        // ---- MLE pass over the partial reads (:3267-3523)
        int prc = S.partial_read_count;
        for (int i = E.tid; i < prc; i += E.nt) {
            E.scr.pflag[i * 2] = 1; E.scr.pflag[i * 2 + 1] = 0;
            if (G == G0) { E.scr.ppos_org[i * 3] = 0; E.scr.ppos_org[i * 3 + 1] = -200; E.scr.ppos_org[i * 3 + 2] = 0; }
        }
        fig_compute_sequence(E, 0, 0);
        for (int x = E.tid; x < S.ncols; x += E.nt) { E.gs[x] = E.scr.cons[x]; for (int j = 0; j < 5; j++) E.scr.ncnt[j * cg + x] = 0; }
        FIG_SYNC();
        // match/mismatch table over the extended columns (overlays the {P,Q} table, which the next computeProbsGap rebuilds)
        fig_build_mle_table(E, G, left, right);          // ends with a barrier
        FIG_TICK(E, 32);
        const double *Ctab = (const double *)E.pq;
        // one read per wave, no workgroup barrier inside the loop (the per-read results are independent)
        if constexpr (LDS) { if (pfast) fig_partial_mle<LDS>(E, gapoffset, nproc, pfl, pml); }
        for (int p = E.wave; p < nproc && !pfast; p += E.nw) {
            unsigned char *rbw = (unsigned char *)E.plb + (E.wave % FIG_PLB_TEAMS) * 256;
            const int len = PR.len[pb + p], flag1 = PR.aux[pb + p];
            { const long long woff = PR.woff[pb + p]; for (int j = E.lane; j < len; j += E.wsz) rbw[j] = (unsigned char)fig_read_code(E.B->packed, woff, len, j); }
            E.rb = rbw;
            int j0 = (flag1 == 1 || flag1 == 4) ? 2 : 0, j1 = len - ((flag1 == 1 || flag1 == 4) ? 0 : 2);
            FigWin w = fig_window_partial(E, PR.pos[pb + p], PR.refpos[pb + p], len, G, gapoffset, 0);
            FigBest best; best.v = -FIG_DBL_MAX; best.o = FIG_NOPOS;
            bool hasn = false;
            { bool n = false; for (int j = E.lane; j < len; j += E.wsz) n = n || rbw[j] > 3; hasn = fig_wave_any(n); }
            fig_cu32p pkr = (fig_cu32p)(fig_uptr(E.B->packed) + fig_u64(PR.woff[pb + p]));
            fig_cdp mtf = (fig_cdp)fig_uptr(E.mt_fwd);
            const int ncolE_u = fig_u(E.ncolE), xoff_u = fig_u(E.xoff), j0u = fig_u(j0), j1u = fig_u(j1);
            for (int o = w.lo + E.lane; o <= w.hi; o += E.wsz) {
                double t = 1;
                // fig_mle_chain through the table: factor = m3[k] on a match, e[k]*T[from][to] otherwise (k = j: forward reads)
                if (hasn) {
                    const double *e_ = E.M->e, *m3_ = E.M->m3;
                    const double *cc = Ctab + o + E.xoff;
                    for (int j = j0; j < j1; j++) {
                        const double c = cc[(long long)rbw[j] * E.ncolE + j];
                        const double f = e_[j] * c;
                        t *= (c < 0 ? m3_[j] : f);
                    }
                } else {            // read bases and {m3, e} pairs through scalar loads
                    const double *cc = Ctab + o + xoff_u;
                    int j = j0u;
                    for (; j + 4 <= j1u; j += 4) {
                        double c[4], kk[8];
#pragma unroll
                        for (int q = 0; q < 4; q++) { const int jj = j + q; const int b = (int)((pkr[jj >> 4] >> ((jj & 15) * 2)) & 3); c[q] = cc[b * ncolE_u + jj]; kk[2 * q] = mtf[2 * jj]; kk[2 * q + 1] = mtf[2 * jj + 1]; }
#pragma unroll
                        for (int q = 0; q < 4; q++) { const double f = kk[2 * q + 1] * c[q]; t *= (c[q] < 0 ? kk[2 * q] : f); }
                    }
                    for (; j < j1u; j++) { const int b = (int)((pkr[j >> 4] >> ((j & 15) * 2)) & 3); const double c = cc[b * ncolE_u + j]; const double f = mtf[2 * j + 1] * c; t *= (c < 0 ? mtf[2 * j] : f); }
                }
                if (t > best.v) { best.v = t; best.o = o; }
                pfl += (unsigned long long)(j1 - j0); pml += (unsigned long long)(j1 - j0);
            }
            best = fig_wave_best(E, best);
            if (E.lane == 0 && p < prc) {
                int o = best.o == FIG_NOPOS ? -left : best.o;
                double temp_log_val = -fig_log10(best.v);
                if (temp_log_val < M.cutoff) {
                    fig_atomic_add_i32(&S.valid_count, 1);
                    E.scr.pflag[p * 2 + 1] = o;
                    if (G == G0) { E.scr.ppos_org[p * 3] = 1; E.scr.ppos_org[p * 3 + 1] = o; E.scr.ppos_org[p * 3 + 2] = len; }
                } else { fig_atomic_add_i32(&S.invalid_count, 1); E.scr.pflag[p * 2] = 0; }
            }
        }
        E.rb = rb_keep;
        FIG_TICK(E, 33);
        E.flops += pfl; E.mle_alg += pml;
        { unsigned long long t = pml; for (int off = 32; off > 0; off >>= 1) t += fig_shfl_down_u64(t, off); E.mle_exec += t; }   // unpruned: executed == credited
        FIG_SYNC();
        FIG_TICK(E, 5);
        int ret_val[2] = {0, 0};
        fig_detect_overlap_par<LDS>(E, E.scr.pflag, 2, G, ret_val, 8);      // all threads; result in thread 0
        if (E.tid == 0) {
            if (ret_val[0] == 300) maxLikelihood += ret_val[0];
            else if (ret_val[0] >= 1 && ret_val[0] < FIG_MAX_READLEN) maxLikelihood += 30 * ret_val[0];
            else if (ret_val[1] == -1) maxLikelihood += -100;
        }
        FIG_SYNC();
    }

    if (M.unmapped == 1) {
        // ---- E-step (:3530-3689): fig_engine_hot.h
        FIG_TICK(E, 4);
        fig_hot_estep_dispatch<LDS>(E, gapoffset);
        FIG_TICK(E, 5);
        fig_dbg_planes(E, nU);
        // ---- consensus + MLE pass (:3694-3914)
        fig_compute_sequence(E, 0, 0);
        for (int x = E.tid; x < S.ncols; x += E.nt) E.gs[x] = E.scr.cons[x];
        FIG_SYNC();
        FIG_TICK(E, 6);
        fig_mle_dispatch<LDS>(E, gapoffset, 0, G, left, right);
        FIG_TICK(E, 7);
        for (int x = E.tid; x < S.ncols; x += E.nt) for (int j = 0; j < 5; j++) E.scr.ncnt[j * cg + x] = (double)E.scr.nci[j * cg + x];
        // ordered likelihood sum (:3852-3862): the terms are fetched wave-wide (one coalesced load per 64 reads) and then
        // added one at a time, in read order, by every lane of the first wave alike (v_readlane broadcast)
        if (E.tid < E.wsz) {
            for (int r0 = 0; r0 < nU; r0 += E.wsz) {
                const int r = r0 + E.lane;
                double term = 0;
                if (r < nU) term = E.scr.accf[r] ? E.scr.maxlv[r] : -50.0;
                const int n = nU - r0 < E.wsz ? nU - r0 : E.wsz;
#ifdef FIG_EMU
                for (int k = 0; k < n; k++) maxLikelihood += term;
#else
                long long bits; memcpy(&bits, &term, 8);
                const int lo32 = (int)(bits & 0xffffffffLL), hi32 = (int)(bits >> 32);
                for (int k = 0; k < n; k++) {
                    const unsigned int l = (unsigned int)__builtin_amdgcn_readlane(lo32, k), h = (unsigned int)__builtin_amdgcn_readlane(hi32, k);
                    const long long b2 = (long long)(((unsigned long long)h << 32) | l);
                    double t; memcpy(&t, &b2, 8);
                    maxLikelihood += t;
                }
#endif
            }
        }
        FIG_SYNC();
        fig_compute_sequence(E, 1, 1);
        // current_str vs previous_str (:3917-3927)
        if (E.tid == 0) { S.ibuf[0] = (S.cons_len == S.prev_len) ? 1 : 0; }
        FIG_SYNC();
        if (S.ibuf[0]) {
            int neq = 0;
            for (int x = E.tid; x < S.cons_len; x += E.nt) if (E.scr.cons[x] != E.scr.prev[x]) neq = 1;
            if (neq) S.ibuf[0] = 0;
        }
        FIG_SYNC();
        int same = S.ibuf[0];
        FIG_SYNC();
        if (same) { if (E.tid == 0) S.comp_count++; }
        else {
            for (int x = E.tid; x < S.cons_len; x += E.nt) E.scr.prev[x] = E.scr.cons[x];
            if (E.tid == 0) { S.prev_len = S.cons_len; S.comp_count = 0; }
        }
        FIG_SYNC();

        if (finalize_flag) {
            if (E.tid == 0) {
                // low-coverage regions (:3935-3977)
                int *region = E.scr.region;
                int region_start = 0, region_count = 0;
                int rl = FIG_MAX_GAP, rr = -FIG_MAX_GAP;
                for (int i = 0; i < G; i++) {
                    int cv = E.scr.cov[i];
                    if (cv < 1 && region_start == 0) { region[region_count] = i; region_start = 1; }
                    else if (cv >= 1 && region_start == 1) {
                        if (i - 1 - region[region_count] >= 10) { region_start = 0; region[region_count + 1] = i - 1; region_count += 2; }
                    }
                    if (i == G - 1 && region_start == 1) {
                        if (i - region[region_count] >= 10) { region_start = 0; region[region_count + 1] = i; region_count += 2; }
                    }
                }
                if (region_count) { rl = region[0]; rr = region[region_count - 1]; S.region_perct = (rr - rl * 1.0) / G; }
                else S.region_perct = 0;
                for (int i = 0; i < nU; i++) {
                    if (E.scr.mark[i] == 1) {
                        if (E.scr.frp[i * 2] >= rl && E.scr.frp[i * 2] + E.scr.frp[i * 2 + 1] - 1 < rr) {
                            maxLikelihood += -50;
                            E.scr.mark[i] = 0;
                            if (G == G0) { E.scr.org[i * 2] = -200; E.scr.org[i * 2 + 1] = 0; }
                            S.valid_count--;
                            E.scr.frp[i * 2] = -200; E.scr.frp[i * 2 + 1] = 0;
                        }
                    }
                }
                maxLikelihood += fig_find_overlap_unmapped(E);
                int condition = M.unmapped && S.comp_count >= 1 && S.region_perct != 0 && ge != S.num_itr - 1;
                if (condition && updateflag) fig_seed_reweight(E);
            }
            FIG_SYNC();
        }
    }
    if (E.tid == 0) S.lik = maxLikelihood;
    FIG_SYNC();
    FIG_TICK(E, 8);
}

// ---------------------------------------------------------------------------------------
// findRegion, Figbird.cpp:4594-4621 (lane 0)
FIG_D int fig_find_region(FigEng &E, int *region) {
    int Nstart = 0, rc = 0, len = E.S->G;
    const unsigned char *con = E.scr.cons;
    int clen = E.S->cons_len;
    for (int i = 0; i < len; i++) {
        int c = i < clen ? con[i] : 255;
        if (c == 4 && Nstart == 0) { region[2 * rc] = i; Nstart = 1; }
        else if (c != 4 && Nstart == 1) { Nstart = 0; region[2 * rc + 1] = i - 1; rc++; }
        if (i == len - 1 && Nstart == 1) { region[2 * rc + 1] = i; rc++; }
    }
    return rc;
}
FIG_D void fig_set_cons(FigEng &E, long long idx, unsigned char c) {
    if (idx >= 0 && idx < E.S->cons_len) E.scr.cons[idx] = c;
}
// recheck_sequence (+findDiscontinous), Figbird.cpp:4623-4743 (lane 0).  pos rows are {pos,len}.
FIG_D int fig_recheck_sequence(FigEng &E, const int *pos) {
    FigState &S = *E.S;
    int *region = E.scr.region;
    int region_count = fig_find_region(E, region);
    int len = S.G, n = E.g->nU, G0 = E.g->G0;
    FigTrip *vec = fig_sort_buf(E, n);
    for (int i = 0; i < n; i++) { vec[i].v[0] = pos[i * 2]; vec[i].v[1] = pos[i * 2 + 1]; vec[i].v[2] = i; }
    fig_std_sort(vec, n);
    int flag = 0;
    for (int i = 0; i < n - 1; i++) {
        if (vec[i].v[0] != -200) {
            int diff = (vec[i].v[0] + vec[i].v[1] - vec[i + 1].v[0]);
            if (diff >= 0 && diff <= 2) { fig_set_cons(E, vec[i].v[0] + vec[i].v[1], 4); flag++; }
        }
    }
    if (flag > 0) region_count = fig_find_region(E, region);
    double reduction_factor = G0 < 400 ? 1 : G0 < 1200 ? 1.5 : 2;
    int readchar = 30;
    if (region_count <= 1) {
        if (region_count == 1) {
            if (S.region_perct_max < .75 || flag > 0) {
                int i, j;
                for (i = region[0] - 1; i >= region[0] - reduction_factor * readchar && i >= 0; i--) fig_set_cons(E, i, 4);
                for (j = region[1] + 1; j <= region[1] + reduction_factor * readchar && j < len; j++) fig_set_cons(E, j, 4);
                if (i < 0 && j == len) return 1;
            }
        }
    } else {
        int start = region[0], end = region[2 * region_count - 1];
        for (int j = start; j < end; j++) fig_set_cons(E, j, 4);
        int i, j;
        for (i = start - 1; i > start - 1 - reduction_factor * readchar && i >= 0; i--) fig_set_cons(E, i, 4);
        for (j = end + 1; j < end + 1 + reduction_factor * readchar && j < len; j++) fig_set_cons(E, j, 4);
        if (i < 0 && j == len) { S.G = G0; fig_ise(E); return 1; }
    }
    return 0;
}

FIG_D void fig_clear_counts(FigEng &E, int a) {          // clear_countsGap, :4583-4592 (lane 0)
    int cg = E.capG;
    if (a > cg) a = cg;
    for (int j = 0; j < a; j++) for (int k = 0; k < 4; k++) E.scr.cnt[k * cg + j] = 0;
}

// check_update, Figbird.cpp:4535-4581 (lane 0); arr = counts of column j (stride capG)
FIG_D int fig_check_update(FigEng &E, int j) {
    int cg = E.capG;
    double max_val = -FIG_DBL_MAX, second_max = -FIG_DBL_MAX;
    int maxp = -FIG_MAX_GAP, sec_p = -FIG_MAX_GAP;
    for (int k = 0; k < 4; k++) {
        double a = E.scr.cnt[k * cg + j];
        if (a > max_val) { second_max = max_val; sec_p = maxp; max_val = a; maxp = k; }
        else if (a >= second_max) { second_max = a; sec_p = k; }
    }
    int diff = (int)(max_val - second_max);
    if (diff >= 2) {
        if (max_val > 3 && second_max > 3) { if (E.scr.tmp[maxp * cg + j] <= E.scr.tmp[sec_p * cg + j]) return maxp; else return sec_p; }
        return 50;
    } else {
        if (max_val >= 1 && second_max >= 1) { if (E.scr.tmp[maxp * cg + j] <= E.scr.tmp[sec_p * cg + j]) return maxp; else return sec_p; }
        return -1;
    }
}

// ---------------------------------------------------------------------------------------
// finalize, Figbird.cpp:4929-5659
template <bool LDS>
FIG_D void fig_finalize(FigEng &E, int gl) {
    FigState &S = *E.S;
    const FigDevModel &M = *E.M;
    int cg = E.capG, G0 = E.g->G0, nU = E.g->nU;
    int gapoffset = gl - G0;
    // gapString uses the window of the LAST candidate (left/right as they stand) and bestString (:4978-4996)
    int gsl_left = S.left, gsl_right = S.right;
    for (int x = E.tid; x < gl; x += E.nt) E.gs[x] = x < S.best_len ? E.scr.best[x] : 4;
    for (int i = E.tid; i < S.partial_read_count; i += E.nt) { E.scr.prf[i * 3] = 0; E.scr.prf[i * 3 + 1] = -200; E.scr.prf[i * 3 + 2] = S.partial_read_len; }
    for (int r = E.tid; r < nU; r += E.nt) { E.scr.fin[r * 2] = -200; E.scr.fin[r * 2 + 1] = 0; }
    // qual_gap is all zero here in the reference too (initialize() cleared it and only finalize adds to it); clear it
    // explicitly because in candidate-parallel mode this workgroup's scratch may never have seen this gap's initialize()
    if (M.partial_flag) for (int x = E.tid; x < cg; x += E.nt) for (int j = 0; j < 5; j++) E.scr.tmp[j * cg + x] = 0;
    {
        long long end_lim = (long long)E.g->alloc_arg;                 // columns relative to left_maxDistance
        long long lim2 = S.end_pos_max - S.left;
        if (lim2 < end_lim) end_lim = lim2;
        if (end_lim > cg) end_lim = cg;
        for (long long x = E.tid; x < end_lim; x += E.nt) for (int j = 0; j < 5; j++) E.scr.cnt[j * cg + x] = 0;
    }
    FIG_SYNC();
    if (E.tid == 0) { S.G = gl; fig_ise(E); for (int q = 0; q < 8; q++) S.ctl[q] = 0; S.ibuf[2] = 0; S.ibuf[3] = 0; }
    FIG_SYNC();
    int G = S.G, left = S.left;
    // lane-0 locals
    int left_right_check[2] = {0, 0};
    int left_start_zero = 0, right_fin_glen = 0, totalCount = 0, discardedCount = 0;
    int unmapped_max_left = 0, unmapped_max_right = 0, leftcount = 0, rightcount = 0;
    int draw_on = E.B->draw_pos != nullptr;

    if (M.unmapped) {
        if (E.tid == 0) { if (E.B->draw_len) E.B->draw_len[(long long)E.g->gapNo * 2] = G; for (int q = 0; q < 8; q++) S.fin_i[q] = 0; }
        FIG_SYNC();
        fig_mle_dispatch<LDS>(E, gapoffset, 1, gl, gsl_left, gsl_right);
        {
            int ncl = gl > cg ? cg : gl;
            for (int x = E.tid; x < ncl; x += E.nt) for (int j = 0; j < 5; j++) E.scr.cnt[j * cg + x] += (double)E.scr.nci[j * cg + x];
        }
        FIG_SYNC();
        if (E.tid == 0) {
            left_right_check[0] = S.fin_i[0]; left_right_check[1] = S.fin_i[1];
            unmapped_max_left = S.fin_i[2]; unmapped_max_right = S.fin_i[3];
            left_start_zero = S.fin_i[4]; right_fin_glen = S.fin_i[5];
            totalCount = S.fin_i[6]; discardedCount = S.fin_i[7];
        }
    }
    if (M.partial_flag) {
        const FigDevReads &PR = E.B->p;
        long long pb = E.g->pBase;
        int np = E.g->nP, nproc = np < FIG_READ_CAP ? np : FIG_READ_CAP;
        if (E.tid == 0 && E.B->draw_len) E.B->draw_len[(long long)E.g->gapNo * 2 + 1] = G;
        if (E.tid == 0 && np > FIG_READ_CAP) totalCount++;              // the 3001st line is counted before the break (:5234-5240)
        for (int p = 0; p < nproc; p++) {
            fig_stage_read(E, PR, pb + p);
            FIG_SYNC();
            int len = PR.len[pb + p], flag1 = PR.aux[pb + p], ref_pos = PR.refpos[pb + p];
            int j0 = (flag1 == 1 || flag1 == 4) ? 2 : 0, j1 = len - ((flag1 == 1 || flag1 == 4) ? 0 : 2);
            FigWin w = fig_window_partial(E, PR.pos[pb + p], ref_pos, len, G, gapoffset, 1);
            FigBest best; best.v = 0; best.o = FIG_NOPOS;
            for (int o = w.lo + E.tid; o <= w.hi; o += E.nt) {
                double t = fig_mle_chain(E, o, len, j0, j1, 0, gl, gsl_left, gsl_right);
                if (t > best.v) { best.v = t; best.o = o; }
                E.flops += (unsigned long long)(j1 - j0);
            }
            best = fig_block_best(E, best);
            int o = best.o == FIG_NOPOS ? -left : best.o;
            double mp = best.o == FIG_NOPOS ? 0.0 : best.v;
            bool acc = (-fig_log10(mp) < M.cutoff) || S.psr_final[0] == p || S.psr_final[1] == p;
            if (acc) {
                const uint8_t *ql = (E.B->qual && E.B->p.qoff) ? E.B->qual + E.B->p.qoff[pb + p] : nullptr;
                for (int j = E.tid; j < len; j += E.nt) {
                    int x = o + j;
                    if (x >= 0 && x < gl) {
                        E.scr.cnt[E.rb[j] * cg + x] += 1;
                        double q = (ql && p < S.partial_read_count && j < M.partial_len) ? M.qtab[ql[j]] : 0.0;
                        E.scr.tmp[E.rb[j] * cg + x] += q;
                    }
                }
            }
            if (E.tid == 0) {
                totalCount++;
                if (acc) {
                    if (o < 0) leftcount++; else rightcount++;
                    if (p < S.partial_read_count) { E.scr.prf[p * 3] = 1; E.scr.prf[p * 3 + 1] = o; E.scr.prf[p * 3 + 2] = len; }
                    if (draw_on) {
                        int rp = ref_pos;
                        if (!((long long)PR.pos[pb + p] < E.g->gapStart)) rp += gapoffset;
                        int placed_pos = (int)(o + E.g->gapStart), nis = -1;
                        if (o < 0) { if (rp != -1) nis = placed_pos - rp + len; }
                        else { if (rp != -1) nis = rp + len - placed_pos; }
                        E.B->draw_pos[E.B->n_ureads + pb + p] = o; E.B->draw_isz[E.B->n_ureads + pb + p] = nis;
                    }
                } else discardedCount++;
            }
            FIG_SYNC();
        }
    }

    // ---- serial tail (:5404-5642)
    if (E.tid == 0) {
        int used_read = totalCount - discardedCount, recompute_flag = 0;
        int Nflag[2] = {-1, -1}, lflag[2] = {-1, -1};
        S.ctl[0] = 0;                                    // 1 => run computeSequence(1,0) before the next serial stage
        S.ctl[1] = used_read; S.ctl[2] = left_right_check[0]; S.ctl[3] = left_right_check[1];
        S.ctl[4] = unmapped_max_left; S.ctl[5] = unmapped_max_right; S.ctl[6] = left_start_zero; S.ctl[7] = right_fin_glen;
        (void)recompute_flag; (void)Nflag; (void)lflag; (void)leftcount; (void)rightcount;
    }
    FIG_SYNC();
    // The tail interleaves lane-0 logic with computeSequence calls (all lanes).  Run it as a small
    // sequence of stages so every lane reaches every barrier.
    int Nflag0 = -1, Nflag1 = -1, lflag0 = -1, lflag1 = -1, recompute_flag = 0;     // lane-0 values
    if (M.unmapped == 1) {
        const int st = 4;
        int need1 = ((S.ctl[4] < st && S.ctl[4] > 0) || (S.ctl[5] < st && S.ctl[5] > 0));
        if (E.tid == 0) {
            if ((S.ctl[4] < 2 * st && S.ctl[4] > 0) || (S.ctl[5] < 2 * st && S.ctl[5] > 0)) { if (S.region_perct_max > .75) S.ctl[1] = 0; }
        }
        FIG_SYNC();
        if (need1) {
            fig_compute_sequence(E, 1, 0);
            if (E.tid == 0) {
                int rc = fig_find_region(E, E.scr.region);
                if (rc >= 1) {
                    if (S.ctl[4] < st && S.ctl[4] > 0) lflag0 = 1;
                    if (S.ctl[5] < st && S.ctl[5] > 0) lflag1 = 1;
                } else if (rc == 0) { S.ctl[1] = 0; S.ctl[4] = S.ctl[5] = -1; }
            }
            FIG_SYNC();
        }
        if (E.tid == 0) {
            if (S.ctl[2] == 0 && S.ctl[3] == 0 && S.ctl[1] != 0) { S.ctl[1] = 0; S.ctl[4] = S.ctl[5] = -1; }
        }
        FIG_SYNC();
        int need2 = ((S.ctl[2] == 0 && S.ctl[6] != 0) || (S.ctl[3] == 0 && S.ctl[7] != 0));
        if (need2) {
            fig_compute_sequence(E, 1, 0);
            if (E.tid == 0) {
                int rc = fig_find_region(E, E.scr.region);
                if (rc >= 1) { if (S.ctl[2] == 0) Nflag0 = 1; if (S.ctl[3] == 0) Nflag1 = 1; }
            }
            FIG_SYNC();
        }
        if (E.tid == 0) {
            if (S.ctl[1] == 0 || (!(S.ctl[2] == 1 && S.ctl[3] == 1))) {
                S.G = G0; fig_ise(E);
                int offset = S.G > gl ? 0 : (gl - S.G);
                fig_clear_counts(E, S.G + offset);
                int do_rc = 0;
                if (!S.ctl[2] && S.ctl[3] && S.ctl[5] >= st) do_rc = 1;
                else if (S.ctl[2] && !S.ctl[3] && S.ctl[4] >= st) do_rc = 1;
                if (do_rc) {                               // recompute2(unmapped_read_pos_arr_org), :4908-4927
                    for (int r = 0; r < nU; r++) {
                        if (E.scr.org[r * 2 + 1] > 0) {
                            int pos = E.scr.org[r * 2];
                            long long idx = E.g->uBase + r;
                            int len = E.B->u.len[idx]; long long woff = E.B->u.woff[idx];
                            for (int j = 0; j < len; j++)
                                if (pos + j >= 0 && pos + j < S.G) E.scr.cnt[fig_read_code(E.B->packed, woff, len, j) * cg + pos + j] += 1;
                        }
                    }
                    recompute_flag = 1;
                }
            }
        }
        FIG_SYNC();
    }
    if (M.partial_flag == 1) {
        if (E.tid == 0) {
            int used_read = S.ctl[1];
            int ret_val[2] = {0, 0};
            int u_flag = 1;
            fig_detect_overlap(E, E.scr.prf, 3, S.G, ret_val, 8);
            int gap_case;
            if ((G0 - gl) > 0 && ret_val[0] > 0) gap_case = 1;
            else if ((G0 - gl) > 0 && ret_val[0] == 0) gap_case = 2;
            else if ((G0 - gl) < 0 && ret_val[0] > 0) gap_case = 3;
            else if ((G0 - gl) < 0 && ret_val[0] == 0) gap_case = 4;
            else gap_case = 5;
            if (used_read < 2 || gap_case == 2 || gap_case == 4) {
                S.G = G0; fig_ise(E);
                int offset = S.G > gl ? 0 : (gl - S.G);
                fig_clear_counts(E, S.G + offset);
                if (used_read < 2 || gap_case == 4) u_flag = 0;
                else {
                    // recompute1(partial_read_pos_arr_org), :4875-4906
                    int np = E.g->nP;
                    for (int p = 0; p < np && p < S.partial_read_count; p++) {
                        if (E.scr.ppos_org[p * 3] == 1) {
                            int pos = E.scr.ppos_org[p * 3 + 1];
                            long long idx = E.g->pBase + p;
                            int len = E.B->p.len[idx]; long long woff = E.B->p.woff[idx];
                            for (int j = 0; j < len; j++)
                                if (pos + j >= 0 && pos + j < S.G) E.scr.cnt[fig_read_code(E.B->packed, woff, len, j) * cg + pos + j] += 1;
                        }
                    }
                    for (int i = 0; i < S.partial_read_count; i++) for (int q = 0; q < 3; q++) E.scr.prf[i * 3 + q] = E.scr.ppos_org[i * 3 + q];
                    fig_detect_overlap(E, E.scr.prf, 3, S.G, ret_val, 8);
                    if (ret_val[1] == -1) { fig_clear_counts(E, S.G + offset); u_flag = 0; }
                }
            }
            if (u_flag == 1 && ret_val[0] == 0 && ret_val[1] == 0) {
                for (int j = 0; j < S.G; j++) {
                    int nz = 0;
                    for (int k = 0; k < 4; k++) if (E.scr.cnt[k * cg + j] > 0) nz++;
                    if (nz) {
                        int uf = fig_check_update(E, j);
                        if (uf != -1) { if (uf != 50) E.scr.cnt[uf * cg + j] += 10; }
                        else for (int k = 0; k < 4; k++) E.scr.cnt[k * cg + j] = 0;
                    }
                }
            }
        }
        FIG_SYNC();
    }
    fig_compute_sequence(E, 1, 0);
    int need3 = (M.unmapped && (S.ctl[2] || S.ctl[3] || S.ctl[1] != 0));
    if (need3) {
        if (E.tid == 0) {
            if (Nflag0 == 1) fig_set_cons(E, 0, 4);
            if (Nflag1 == 1) fig_set_cons(E, S.G - 1, 4);
            if (lflag0 == 1) fig_set_cons(E, 0, 4);
            if (lflag1 == 1) fig_set_cons(E, S.G - 1, 4);
            int clear_val = fig_recheck_sequence(E, recompute_flag == 0 ? E.scr.fin : E.scr.org);
            S.bc_i = clear_val;
            if (clear_val == 1) {
                S.G = G0; fig_ise(E);
                int offset = S.G > gl ? 0 : (gl - S.G);
                fig_clear_counts(E, S.G + offset);
            }
        }
        FIG_SYNC();
        if (S.bc_i == 1) fig_compute_sequence(E, 1, 0);
    }
    FIG_SYNC();
}

// ---------------------------------------------------------------------------------------
// run, Figbird.cpp:5913-5965.  Result (valid_count or likelihood) in S.bc_d; S.bc_j = 1 if side_limit < 10.
template <bool LDS>
FIG_D void fig_run(FigEng &E, int gaplen, int finalize_flag, int c) {
    FigState &S = *E.S;
    if (E.tid == 0) { S.left = E.M->D; S.right = E.M->D; }
    FIG_SYNC();
    fig_initialize(E, gaplen, 1);
    if (S.side_limit < 10) { if (E.tid == 0) { S.bc_d = 10000; S.bc_j = 1; } FIG_SYNC(); return; }
    if (E.tid == 0) { S.comp_count = 0; S.lik = 0; }
    FIG_SYNC();
    for (int p = 0; p < S.num_itr; p++) {
        if (E.tid == 0) { S.valid_count = 0; S.invalid_count = 0; }
        FIG_SYNC();
        fig_place_reads<LDS>(E, p, finalize_flag, gaplen - E.g->G0, 0);
        fig_compute_probs(E);
        FIG_SYNC();
        if (E.M->unmapped) { if (S.comp_count >= 5) break; }
        else { if (p == 2) break; }
    }
    if (E.tid == 0) { S.region_perct_max = S.region_perct; S.bc_d = (c == 0) ? (double)S.valid_count : S.lik; S.bc_j = 0; }
    FIG_SYNC();
}

// checkGapReads, Figbird.cpp:6121-6153
template <bool LDS>
FIG_D int fig_check_gap_reads(FigEng &E, int org) {
    FigState &S = *E.S;
    if (org < 30) {
        int step = org < 15 ? 10 : 20;
        for (int i = 0; i < 80; i += step) {
            fig_run<LDS>(E, i, 1, 1);
            if (S.bc_j == 1) return -2;
            if (S.valid_count > 3) return -1;
        }
    } else {
        for (int k = 0; k < 4; k++) {
            int gap = k == 0 ? org / 2 : org * k;
            fig_run<LDS>(E, gap, 1, 1);
            if (S.bc_j == 1) return -2;
            if (S.valid_count >= 3) return -1;
        }
    }
    return 1;
}

FIG_D void fig_dbg_cand(FigEng &E, int gapEstimate, int iters, int valid, double lik) {
    const FigDevBatch &B = *E.B;
    if (!B.dbg_n_cand) return;
    long long g = E.g->gapNo;
    int n = B.dbg_n_cand[g];
    if (n < B.dbg_max_cand) {
        long long o = g * B.dbg_max_cand + n;
        B.dbg_cand_i[o * 3] = gapEstimate; B.dbg_cand_i[o * 3 + 1] = iters; B.dbg_cand_i[o * 3 + 2] = valid;
        B.dbg_cand_lik[o] = lik;
    }
    B.dbg_n_cand[g] = n + 1;
}

// run(originalGap,...,1,0); computeSequence(0,0); strcpy(<dst>,concensus)  -- the recurring fallback (:6411-6413 etc.)
template <bool LDS>
FIG_D void fig_run_original(FigEng &E, unsigned char *dst, int *dlen) {
    fig_run<LDS>(E, E.g->G0, 1, 0);
    fig_compute_sequence(E, 0, 0);
    fig_copy_str(E, dst, dlen, E.scr.cons, E.S->cons_len);
}

#include "fig_engine_sched.h"
