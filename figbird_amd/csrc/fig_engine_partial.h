// fig_engine_partial.h -- the partial-mode (frag library) hot loops of placeReads for gfx950: the E-step over the soft-clipped
// reads (Figbird.cpp:3082-3264) and their MLE pass (:3267-3523).  Included by fig_engine_core.h after fig_engine_hot.h.
//
// A partial read is anchored in one flank, so it has at most len-1 placements (left side: o in [-(len-1), -1], right side:
// o in [G-len+1, G-1], cut by the insert-size window when the mate is mapped) and a gap holds a few dozen of them: a
// placeReads call is ~2 MFLOP.  What it costs is therefore orchestration, not arithmetic, and the form below is built around
// that (round 4; the round-1 form -- one read per wave and chunk, a workgroup barrier pair per chunk of `nw` reads, lanes =
// columns for the ordered pile-up, every table and row access through generic pointers -- stays as the fallback for reads
// with an N base, gaps clipped by a contig start and the classes whose table is not in LDS):
//
//   phase A  every wave takes reads p = wave, wave + nw, ... of a super-chunk of up to FIG_PT_ROWS reads; lanes = placements,
//            TWO per lane (o and o + 64: a 101-bp read's 100 placements are one round), the pair chain of the unmapped hot
//            path (fig_eblk_*: read bases and {1-e, e} pairs through scalar loads, one ds_read_b128 per step and placement,
//            8-step blocks, a lane's two placements on one address register); the clipped bases (clip_thresh = 2,
//            :3117-3118) are left out by the block's step range; weights w = pow(10, log p) go to the read's LDS row
//            [0 | w(lo) .. w(hi) | 0], the read's maximum through one DPP reduction;
//   barrier
//   phase B  wave b owns base b and adds, for every read of the super-chunk in file order and every position j of base b
//            in it (the packed record's position lists, descending j = ascending placement), row[x - j] to its register
//            accumulators (two column tiles of a read side by side), lanes = columns x: the reference's (read, placement) order per (column, base), no atomics.  Only
//            the columns a partial read can reach have accumulators: x < 64 TS from the left and x >= G - 64 TS from the
//            right (64 TS >= L - 1); the index is clamped into the row's zero slots instead of padding the row by a tile;
//   barrier (the rows are reused by the next super-chunk)
//
// and after the last super-chunk the per-read maxima are added in file order (:3258-3261).  The MLE pass runs the same
// two-placements-per-lane chain on the match/mismatch table (fig_mblk_*), one read per wave with no barrier.
#ifndef FIG_ENGINE_PARTIAL_H
#define FIG_ENGINE_PARTIAL_H

#define FIG_PT_ROWS 16               // reads per super-chunk (FigState::tm_* hold 16 entries)

// Typed views (no FLAT instruction in the per-read code): the batch's read arrays are read-only and wave-uniformly indexed ->
// constant address space (s_load); the slab arrays are global; FigState is LDS.
#ifdef FIG_EMU
typedef const int32_t *fig_ci32p;
typedef const int64_t *fig_ci64p;
typedef int *fig_gip;
typedef FigState *fig_plsp;
#else
typedef const int32_t __attribute__((address_space(4))) *fig_ci32p;
typedef const int64_t __attribute__((address_space(4))) *fig_ci64p;
typedef int __attribute__((address_space(1))) *fig_gip;
typedef FigState __attribute__((address_space(3))) *fig_plsp;
#endif
struct FigPartU {                    // wave-uniform copies of what the partial hot loops need
    fig_ci32p len, aux, pos, refpos; fig_ci64p woff; fig_cu32p packed_c; fig_gcu32p packed_g;
    fig_gdp maxlv, cnt; fig_gip pflag, ppos_org;
    long long pb, gapStart;
    int G, G0, cg, left, ncolE, xoff, wave, nw, wsz, L, Tmin, Tmax, cutoff, prc;
};
FIG_D FigPartU fig_part_uniforms(const FigEng &E) {
    FigPartU U;
    const FigDevBatch *B = fig_uptr(E.B);
    const FigDevModel *M = fig_uptr(E.M);
    const FigDevGap *g = fig_uptr(E.g);
    U.len = (fig_ci32p)fig_uptr(B->p.len); U.aux = (fig_ci32p)fig_uptr(B->p.aux); U.pos = (fig_ci32p)fig_uptr(B->p.pos); U.refpos = (fig_ci32p)fig_uptr(B->p.refpos);
    U.woff = (fig_ci64p)fig_uptr(B->p.woff); U.packed_c = (fig_cu32p)fig_uptr(B->packed); U.packed_g = (fig_gcu32p)fig_uptr(B->packed);
    U.maxlv = (fig_gdp)fig_uptr(E.scr.maxlv); U.cnt = (fig_gdp)fig_uptr(E.scr.cnt); U.pflag = (fig_gip)fig_uptr(E.scr.pflag); U.ppos_org = (fig_gip)fig_uptr(E.scr.ppos_org);
    U.pb = fig_u64(g->pBase); U.gapStart = fig_u64(g->gapStart);
    U.G = fig_u(E.S->G); U.G0 = fig_u(g->G0); U.cg = fig_u(E.capG); U.left = fig_u(E.S->left); U.ncolE = fig_u(E.ncolE); U.xoff = fig_u(E.xoff);
    U.wave = fig_u(E.wave); U.nw = fig_u(E.nw); U.wsz = E.wsz; U.L = fig_u(M->L); U.Tmin = fig_u(M->Tmin); U.Tmax = fig_u(M->Tmax); U.cutoff = fig_u(M->cutoff);
    U.prc = fig_u(E.S->partial_read_count);
    return U;
}
// fig_window_partial on uniforms (shift_always = 0: the placeReads form)
FIG_D void fig_window_partial_u(const FigPartU &U, int pos1, int ref_pos, int len, int gapoffset, int &lo, int &hi) {
    if ((long long)pos1 < U.gapStart) {
        lo = -(len - 1); hi = -1;
        if (ref_pos != -1) {
            const int tis0 = (int)(U.gapStart - ref_pos + len);
            const int a = U.Tmin - tis0, b = U.Tmax - tis0;
            if (a > lo) lo = a;
            if (b < hi) hi = b;
        }
    } else {
        int rp = ref_pos;
        if (rp != -1) rp += gapoffset;
        lo = U.G - len + 1; hi = U.G - 1;
        if (rp != -1) {
            const int tis0 = (int)(rp - U.gapStart + len);
            const int a = tis0 - U.Tmax, b = tis0 - U.Tmin;
            if (a > lo) lo = a;
            if (b < hi) hi = b;
        }
    }
}

// The second placement of a lane is always the first + 64 (o and o + wsz), so both loads of a step share one address register and
// the second takes the instruction's constant offset (fig_eblk_load's DX form); a lane whose second placement lies outside the
// window reads at most 64 columns past it -- inside the table area, result ignored.  (One emulated lane: wsz = 1, plain form.)
#ifdef FIG_EMU
#define FIG_PDX 0
#else
#define FIG_PDX 64
#endif

// steps [a, b) of an 8-step block (wave-uniform bounds)
FIG_D void fig_eblk_compute_r(const FigEBlk &B, int a, int b, double &pa, double &pb) {
#pragma unroll
    for (int jj = 0; jj < 8; jj++) {
        if (jj >= a && jj < b) {
            pa *= (B.va[jj].p * B.kk[2 * jj] + B.kk[2 * jj + 1] * B.va[jj].q);
            pb *= (B.vb[jj].p * B.kk[2 * jj] + B.kk[2 * jj + 1] * B.vb[jj].q);
        }
    }
}
FIG_D void fig_mblk_compute_r(const FigMBlk &B, int a, int b, double &qa, double &qb) {
#pragma unroll
    for (int jj = 0; jj < 8; jj++) {
        if (jj >= a && jj < b) {
            const double fa = B.kk[2 * jj + 1] * B.va[jj], fb = B.kk[2 * jj + 1] * B.vb[jj];
            qa *= (B.va[jj] < 0 ? B.kk[2 * jj] : fa);
            qb *= (B.vb[jj] < 0 ? B.kk[2 * jj] : fb);
        }
    }
}

// E-step products of two placements over the read positions [j0, j1) (Figbird.cpp:3142-3167) are fig_pchain_e2 below: xa / xb =
// extended column of the placement's base 0; blocks are aligned to multiples of 8 read positions.

// ---- t = ln p and w = pow(10, t) from the ROUNDED t (Figbird.cpp:3169-3179 takes pow(10, log(tempProb))), N values side
// by side.  ln: x = m 2^e with m in [sqrt(1/2), sqrt(2)), s = (m-1)/(m+1) through v_rcp_f64 + one Newton step with the
// quotient's error recovered, ln m = 2 s + s z q(z) (atanh series), ln x = e ln 2 + ln m assembled from hi/lo parts (the log
// of fig_weights_n without its log10(e) factor).  10^t = 2^k exp(r): t ln 10 as a two-term product, k = rint(t log2 10), r
// reduced with a two-part ln 2, exp(r) by the Taylor polynomial to r^13.  ~75 FP64 operations against ~500 of the library's
// log + pow.  tools/ubench/pweights_check.c runs the same operations on the host: against glibc on 2e7 arguments over
// [2^-1070, 1], ln p is equal on 99.56 % (max 1 ulp) and 10^t is within 1 ulp wherever ln p is equal (a 1-ulp difference
// in t moves w by up to ~10^3 ulp whatever computes it: pow's sensitivity).  Exact for p = 0 (-inf, 0) and p = 1 (0, 1).
template <int N>
FIG_D void fig_pweights(const double (&x)[N], double (&t)[N], double (&w)[N]) {
#ifdef FIG_EMU
    for (int i = 0; i < N; i++) { t[i] = fig_log(x[i]); w[i] = fig_pow10(t[i]); }
#else
    double m[N], a[N], b[N], r[N], s[N], z[N], q[N], s_lo[N], t3[N], lm_lo[N], ed[N], r_hi[N], sum[N], err[N];
    int e[N];
#pragma unroll
    for (int i = 0; i < N; i++) { m[i] = __builtin_amdgcn_frexp_mant(x[i]); e[i] = __builtin_amdgcn_frexp_exp(x[i]); }
#pragma unroll
    for (int i = 0; i < N; i++) { const bool lowhalf = m[i] < 0.70710678118654752440; m[i] = lowhalf ? m[i] + m[i] : m[i]; e[i] = lowhalf ? e[i] - 1 : e[i]; }
#pragma unroll
    for (int i = 0; i < N; i++) { a[i] = m[i] - 1.0; b[i] = m[i] + 1.0; r[i] = __builtin_amdgcn_rcp(b[i]); }
#pragma unroll
    for (int i = 0; i < N; i++) { const double e0 = __builtin_fma(-b[i], r[i], 1.0); r[i] = __builtin_fma(r[i], e0, r[i]); }
#pragma unroll
    for (int i = 0; i < N; i++) { s[i] = a[i] * r[i]; z[i] = s[i] * s[i]; q[i] = 2.0 / 23.0; }
#define FIG_PW_H(c) _Pragma("unroll") for (int i = 0; i < N; i++) q[i] = __builtin_fma(q[i], z[i], c);
    FIG_PW_H(2.0 / 21.0) FIG_PW_H(2.0 / 19.0) FIG_PW_H(2.0 / 17.0) FIG_PW_H(2.0 / 15.0) FIG_PW_H(2.0 / 13.0)
    FIG_PW_H(2.0 / 11.0) FIG_PW_H(2.0 / 9.0) FIG_PW_H(2.0 / 7.0) FIG_PW_H(2.0 / 5.0) FIG_PW_H(2.0 / 3.0)
#undef FIG_PW_H
    const double LN2_HI = 0x1.62e42fee00000p-1, LN2_LO = 0x1.a39ef35793c76p-33;       // hi holds 32 bits: e * LN2_HI and k * LN2_HI are exact
#pragma unroll
    for (int i = 0; i < N; i++) { s_lo[i] = __builtin_fma(-s[i], b[i], a[i]) * r[i]; t3[i] = s[i] * z[i] * q[i]; }
#pragma unroll
    for (int i = 0; i < N; i++) { const double lm_hi = 2.0 * s[i]; lm_lo[i] = __builtin_fma(2.0, s_lo[i], t3[i]); ed[i] = (double)e[i]; r_hi[i] = ed[i] * LN2_HI;
                                  sum[i] = r_hi[i] + lm_hi; err[i] = (r_hi[i] - sum[i]) + lm_hi; if (e[i] == 0) { sum[i] = lm_hi; err[i] = 0.0; } }
#pragma unroll
    for (int i = 0; i < N; i++) { const double res = sum[i] + (err[i] + __builtin_fma(ed[i], LN2_LO, lm_lo[i])); t[i] = x[i] == 0.0 ? -__builtin_inf() : res; }
    const double C_HI = 0x1.26bb1bbb55516p+1, C_LO = -0x1.f48ad494ea3e9p-53, L2E = 0x1.71547652b82fep+0;      // ln 10 in two parts, 1 / ln 2
    double u_hi[N], u_lo[N], kd[N], rh[N], rl[N], rr[N], g[N];
#pragma unroll
    for (int i = 0; i < N; i++) { u_hi[i] = t[i] * C_HI; u_lo[i] = __builtin_fma(t[i], C_HI, -u_hi[i]) + t[i] * C_LO; kd[i] = __builtin_rint(u_hi[i] * L2E); }
#pragma unroll
    for (int i = 0; i < N; i++) { rh[i] = __builtin_fma(-kd[i], LN2_HI, u_hi[i]); rl[i] = __builtin_fma(-kd[i], LN2_LO, u_lo[i]); rr[i] = rh[i] + rl[i]; g[i] = 1.0 / 6227020800.0; }
#define FIG_PW_H(c) _Pragma("unroll") for (int i = 0; i < N; i++) g[i] = __builtin_fma(g[i], rr[i], c);
    FIG_PW_H(1.0 / 479001600.0) FIG_PW_H(1.0 / 39916800.0) FIG_PW_H(1.0 / 3628800.0) FIG_PW_H(1.0 / 362880.0) FIG_PW_H(1.0 / 40320.0) FIG_PW_H(1.0 / 5040.0)
    FIG_PW_H(1.0 / 720.0) FIG_PW_H(1.0 / 120.0) FIG_PW_H(1.0 / 24.0) FIG_PW_H(1.0 / 6.0) FIG_PW_H(0.5)
#undef FIG_PW_H
#pragma unroll
    for (int i = 0; i < N; i++) {
        const double tt = rr[i] * rr[i] * g[i];
        const double ss = 1.0 + rh[i], ee = (1.0 - ss) + rh[i];
        const double res = __builtin_ldexp(ss + ((ee + rl[i]) + tt), (int)kd[i]);
        w[i] = t[i] >= -330.0 ? res : 0.0;
    }
#endif
}

template <bool LDS>
FIG_D void fig_pchain_e2(const FigPQ *PQ, int ncolE, fig_cu32p pk, fig_cdp kt, int j0, int j1, int xa, int xb, double &pa, double &pb) {
    if (j1 <= j0) return;
    const int b0 = j0 >> 3, b1 = (j1 - 1) >> 3;
    // One 8-step block at a time: first block (may start inside it), full blocks without a branch per step, last block (may
    // end inside it).  The two-register-set form of the unmapped pair chain (loads of block i + 1 ahead of the arithmetic of
    // block i) buys nothing here -- with one wave per SIMD and workgroup the chain is bound by the ~135 instructions a block
    // issues, two waves sharing the SIMD's vector unit (3 841 against 3 898 gaps/s) -- and costs 64 more VGPRs.
    for (int bi = b0; bi <= b1; bi++) {
        FigEBlk A;
        fig_eblk_load<LDS, FIG_PDX>(A, PQ, ncolE, pk, kt, bi, xa, xb);
        if (bi == b0 || bi == b1) fig_eblk_compute_r(A, bi == b0 ? j0 - 8 * b0 : 0, bi == b1 ? j1 - 8 * b1 : 8, pa, pb);
        else fig_eblk_compute(A, pa, pb);
    }
}

// MLE products of two placements over [j0, j1) (Figbird.cpp:3383-3407): C[to][x] = -1 on a match, T[from][to] otherwise.
template <bool LDS>
FIG_D void fig_pchain_m2(const double *C, int ncolE, fig_cu32p pk, fig_cdp mt, int j0, int j1, int xa, int xb, double &qa, double &qb) {
    if (j1 <= j0) return;
    const int b0 = j0 >> 3, b1 = (j1 - 1) >> 3;
    for (int bi = b0; bi <= b1; bi++) {
        FigMBlk A;
        fig_mblk_load<LDS, FIG_PDX>(A, C, ncolE, pk, mt, bi, xa, xb);
        if (bi == b0 || bi == b1) fig_mblk_compute_r(A, bi == b0 ? j0 - 8 * b0 : 0, bi == b1 ? j1 - 8 * b1 : 8, qa, qb);
        else fig_mblk_compute(A, qa, qb);
    }
}

// Does the fast form apply to this placeReads call?  (Wave-uniform; the same answer in every thread.)
template <bool LDS>
FIG_D bool fig_partial_fast(const FigEng &E) {
    if (!LDS || !E.pq_lds || E.tiles > 0) return false;
    if (E.g->pad & 1) return false;                               // a partial read of this gap holds a base outside ACGT (fig_pack.h)
    if (E.S->left < E.xoff) return false;                         // placements hang over the contig start (`index<0`, :3155)
#ifndef FIG_EMU
    if (E.nw < 4 || (E.nw & 3)) return false;                     // phase B: one wave per base
#endif
    const int rstride = ((E.M->L + 1 + 7) & ~7) + 32;             // [0 | up to L-1 weights | 0] + the read's position lists
    if ((long long)E.nteams * E.Wcap < (long long)rstride) return false;
    const int ts = (E.M->L - 1 + 63) >> 6;
    if (ts > 4) return false;
    return true;
}

// median of (x, 0, hi) (v_med3_i32): clamps x into [0, hi] in one instruction (hi wave-uniform: one SGPR on the constant bus)
FIG_D int fig_clamp0_i32(int x, int hi) {
#ifdef FIG_EMU
    return x < 0 ? 0 : (x > hi ? hi : x);
#else
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "s"(hi));
    return r;
#endif
}

// Position lists of one read's packed record: word k of the record behind the N mask ({cntA..cntT}, {cntN}, lists), fetched
// one word per lane into `plv`.
#ifndef FIG_EMU
#define FIG_PT_LW(k) ((uint32_t)__builtin_amdgcn_readlane((int)plv, (k)))
#endif

// sub-phase timers of the diagnostic build (slots the unmapped path leaves unused in a partial-mode pass)
#ifdef FIG_PROF
#define FIG_PT0() unsigned long long _fig_pt = __builtin_readcyclecounter()
#define FIG_PTICK(E, slot) do { unsigned long long _n = __builtin_readcyclecounter(); (E).prof[slot] += _n - _fig_pt; _fig_pt = _n; } while (0)
#else
#define FIG_PT0() do { } while (0)
#define FIG_PTICK(E, slot) do { } while (0)
#endif

// ---- E-step (:3082-3264).  maxLikelihood is meaningful in thread 0.  TS = 64-column tiles per side (64 TS >= L - 1).
template <bool LDS, int TS>
FIG_NOINLINE FIG_D void fig_partial_estep_t(FigEng &E, int gapoffset_, int nproc_, double &maxLikelihood_out, unsigned long long &pfl_out) {
    const FigPartU U = fig_part_uniforms(E);
    const fig_plsp SL = (fig_plsp)E.S;
    const int gapoffset = fig_u(gapoffset_), nproc = fig_u(nproc_);
    const long long pb = U.pb;
    const int G = U.G, cg = U.cg;
    const int wave = U.wave, lane = E.lane, nw = U.nw, wsz = U.wsz;
    const FigPQ *PQ = fig_pq_ptr<LDS>(E);
    double *W = fig_w_ptr<LDS>(E);
    const int wstride = (U.L + 1 + 7) & ~7;                        // the weights of a row: [0 | up to L-1 weights | 0]
    const int rstride = wstride + 32;                             // + the read's position-list record (64 dwords), staged by the wave that owns the read
    int K = (int)(((long long)fig_u(E.nteams) * fig_u(E.Wcap)) / rstride);
    if (K > FIG_PT_ROWS) K = FIG_PT_ROWS;
    const fig_cu32p packed_c = U.packed_c;
    const fig_cdp ktf = (fig_cdp)fig_uptr(E.kt_fwd);
    const int ncolE = U.ncolE, xoff = U.xoff;
    unsigned long long pfl = 0;
    double maxLikelihood = maxLikelihood_out;
    const bool first_wave = E.tid < wsz;
    FIG_T0(E);
    // column tiles with accumulators: tile t < TS covers x = 64 t + lane; tile TS + t covers x = xr0 + 64 t + lane.  When all
    // columns fit in 2 TS tiles (modeA) the tiles are simply consecutive.
    const bool modeA = G <= 128 * TS;
    const int xr0 = modeA ? 64 * TS : G - 64 * TS;
    double acc[2 * TS];
#pragma unroll
    for (int t = 0; t < 2 * TS; t++) acc[t] = 0.0;
    const int myb = wave & 3, grp = wave >> 2, ngrp = nw >> 2;    // base this wave owns in phase B; tiles are dealt over the wave groups
    (void)cg;
    for (int c0 = 0; c0 < nproc; c0 += K) {
        const int nr = nproc - c0 < K ? nproc - c0 : K;
        // ---- phase A
        for (int t = wave; t < nr; t += nw) {
            FIG_PT0();
            const int p = c0 + t;
            const int len = U.len[pb + p], flag1 = U.aux[pb + p];
            const long long woff = U.woff[pb + p];
            const bool leftside = flag1 == 1 || flag1 == 4;
            const int j0 = leftside ? 2 : 0, j1 = len - (leftside ? 0 : 2);
            int lo, hi;
            fig_window_partial_u(U, U.pos[pb + p], U.refpos[pb + p], len, gapoffset, lo, hi);
            double *row = W + (long long)t * rstride;
            const fig_cu32p pk = packed_c + woff;
            FigBest best; best.v = -FIG_DBL_MAX; best.o = FIG_NOPOS;
            const int n = hi - lo + 1;
            if (n > 0) {
#ifndef FIG_EMU
                {   // the read's counts + position lists for phase B (the load's latency hides behind the chains)
                    const int ndw = 2 + ((len + 3) >> 2) + 4;
                    const uint32_t pv = lane < ndw ? U.packed_g[woff + ((len + 15) >> 4) + ((len + 31) >> 5) + lane] : 0u;
                    ((uint32_t *)(row + wstride))[lane] = pv;
                }
#endif
                if (lane == 0) { row[0] = 0.0; row[n + 1] = 0.0; }
                FIG_PTICK(E, 9);
                for (int ob = lo; ob <= hi; ob += 2 * wsz) {
                    const int oa = ob + lane, oc = oa + wsz;
                    const bool va = oa <= hi, vc = oc <= hi;
                    double pa = 1.0, pc = 1.0;
                    fig_pchain_e2<LDS>(PQ, ncolE, pk, ktf, j0, j1, (va ? oa : lo) + xoff, (vc ? oc : lo) + xoff, pa, pc);
                    FIG_PTICK(E, 10);
                    double px[2] = {pa, pc}, tx[2], wx[2];
                    fig_pweights<2>(px, tx, wx);
                    const double ta = tx[0], tc = tx[1], wa = wx[0], wc = wx[1];
                    if (va) { row[1 + oa - lo] = wa; if (ta > best.v) { best.v = ta; best.o = oa; } pfl += 4ULL * (unsigned long long)(j1 - j0) + (unsigned long long)fig_ovl(oa, len, G); }
                    if (vc) { row[1 + oc - lo] = wc; if (tc > best.v) { best.v = tc; best.o = oc; } pfl += 4ULL * (unsigned long long)(j1 - j0) + (unsigned long long)fig_ovl(oc, len, G); }
                    FIG_PTICK(E, 22);
                }
                best = fig_wave_best(E, best);
                FIG_PTICK(E, 23);
            }
            if (lane == 0) {
                SL->tm_lo[t] = lo; SL->tm_hi[t] = hi; SL->tm_len[t] = len;
                U.maxlv[p] = best.o != FIG_NOPOS ? best.v : 0.0;           // (also parity plane (ii))
                U.pflag[p * 2] = best.o != FIG_NOPOS ? 1 : 0;              // valid maximum (pflag is re-initialised before the MLE pass)
            }
        }
        FIG_TICK(E, 27);
        FIG_SYNC();
        FIG_TICK(E, 28);
        // ---- phase B
#ifdef FIG_EMU
        // one emulated lane: the reference's loop order, straight (Figbird.cpp:3181-3187)
        for (int t = 0; t < nr; t++) {
            const int p = c0 + t, lo = SL->tm_lo[t], hi = SL->tm_hi[t], len = SL->tm_len[t];
            const double *row = W + (long long)t * rstride;
            const long long woff = U.woff[pb + p];
            for (int o = lo; o <= hi; o++)
                for (int j = 0; j < len; j++) {
                    const int x = o + j;
                    if (x >= 0 && x < G) U.cnt[fig_read_code(E.B->packed, woff, len, j) * cg + x] += row[1 + o - lo];
                }
        }
        (void)acc; (void)myb; (void)grp; (void)ngrp; (void)xr0; (void)modeA;
#else
        {
            for (int t = 0; t < nr; t++) {
                FIG_PT0();
                const int lo = fig_u(SL->tm_lo[t]), hi = fig_u(SL->tm_hi[t]), len = fig_u(SL->tm_len[t]);
                const int n = hi - lo + 1;
                if (n <= 0) continue;
                // columns this read reaches: [max(0, lo), min(G - 1, hi + len - 1)] -> mask of the tiles that overlap them
                const int xa = lo > 0 ? lo : 0, xz = hi + len - 1 < G - 1 ? hi + len - 1 : G - 1;
                if (xz < xa) continue;
                const uint32_t plv = ((const uint32_t *)(W + (long long)t * rstride + wstride))[lane];
                const uint32_t cw0 = FIG_PT_LW(0);
                const int nA = (int)(cw0 & 255), nC = (int)((cw0 >> 8) & 255), nG = (int)((cw0 >> 16) & 255), nT_ = (int)(cw0 >> 24);
                const int dC = 2 + ((nA + 3) >> 2), dG = dC + ((nC + 3) >> 2), dT = dG + ((nG + 3) >> 2);
                const int nb = myb == 0 ? nA : myb == 1 ? nC : myb == 2 ? nG : nT_;
                const int d0 = myb == 0 ? 2 : myb == 1 ? dC : myb == 2 ? dG : dT;
                // one pass over the base's positions per tile the read reaches (at most three; two when the accumulators are
                // split into a left and a right region); per (position, tile): one subtract, one v_med3 that clamps the row index
                // into [zero slot 0, zero slot n + 1], the ds_read_b64 and the add
                const double *row0 = W + (long long)t * rstride;               // zero slot 0; weights at 1 .. n; zero slot n + 1
                FIG_PTICK(E, 24);
                // one tile: acc[Q] += row0[clamp(x - j)] over the base's positions, in list order
#define FIG_PB_TILE1(Q, XB) do { \
                    const int xl1 = (XB) + lane - lo + 1; \
                    double a = acc[Q]; \
                    int k4 = 0; \
                    for (; 4 * k4 + 4 <= nb; k4++) { \
                        const uint32_t s4 = FIG_PT_LW(d0 + k4); \
                        const int i0 = fig_clamp0_i32(xl1 - (int)(s4 & 255), n + 1), i1 = fig_clamp0_i32(xl1 - (int)((s4 >> 8) & 255), n + 1); \
                        const int i2 = fig_clamp0_i32(xl1 - (int)((s4 >> 16) & 255), n + 1), i3 = fig_clamp0_i32(xl1 - (int)(s4 >> 24), n + 1); \
                        const double v0 = row0[i0], v1 = row0[i1], v2 = row0[i2], v3 = row0[i3]; \
                        a += v0; a += v1; a += v2; a += v3; \
                    } \
                    const int rem = nb - 4 * k4; \
                    if (rem > 0) { \
                        const uint32_t s4 = FIG_PT_LW(d0 + k4); \
                        for (int k = 0; k < rem; k++) a += row0[fig_clamp0_i32(xl1 - (int)((s4 >> (8 * k)) & 255), n + 1)]; \
                    } \
                    acc[Q] = a; } while (0)
                // two tiles side by side: one pass over the position list, eight loads in flight, two independent add chains (one
                // tile at a time a lone wave sits out a load round trip and four dependent adds per group of four positions).
                // Written as macros on acc[constant]: through lambdas the accumulators lose their registers.  (Issuing group g + 1's
                // loads ahead of group g's adds over two register sets -- the compiler then interleaves them with counted waits --
                // is SLOWER: 3 431 against 3 841 gaps/s.)
#define FIG_PB_TILE2(QA, QB, XBA, XBB) do { \
                    const int xA = (XBA) + lane - lo + 1, xB = (XBB) + lane - lo + 1; \
                    double a = acc[QA], b2 = acc[QB]; \
                    int k4 = 0; \
                    for (; 4 * k4 + 4 <= nb; k4++) { \
                        const uint32_t s4 = FIG_PT_LW(d0 + k4); \
                        const int j0_ = (int)(s4 & 255), j1_ = (int)((s4 >> 8) & 255), j2_ = (int)((s4 >> 16) & 255), j3_ = (int)(s4 >> 24); \
                        const double u0 = row0[fig_clamp0_i32(xA - j0_, n + 1)], w0 = row0[fig_clamp0_i32(xB - j0_, n + 1)]; \
                        const double u1 = row0[fig_clamp0_i32(xA - j1_, n + 1)], w1 = row0[fig_clamp0_i32(xB - j1_, n + 1)]; \
                        const double u2 = row0[fig_clamp0_i32(xA - j2_, n + 1)], w2 = row0[fig_clamp0_i32(xB - j2_, n + 1)]; \
                        const double u3 = row0[fig_clamp0_i32(xA - j3_, n + 1)], w3 = row0[fig_clamp0_i32(xB - j3_, n + 1)]; \
                        a += u0; b2 += w0; a += u1; b2 += w1; a += u2; b2 += w2; a += u3; b2 += w3; \
                    } \
                    const int rem = nb - 4 * k4; \
                    if (rem > 0) { \
                        const uint32_t s4 = FIG_PT_LW(d0 + k4); \
                        for (int k = 0; k < rem; k++) { \
                            const int j_ = (int)((s4 >> (8 * k)) & 255); \
                            const double u = row0[fig_clamp0_i32(xA - j_, n + 1)], w = row0[fig_clamp0_i32(xB - j_, n + 1)]; \
                            a += u; b2 += w; \
                        } \
                    } \
                    acc[QA] = a; acc[QB] = b2; } while (0)
                if (ngrp == 1) {
#pragma unroll
                    for (int q = 0; q < 2 * TS; q += 2) {
                        const int xbA = q < TS ? 64 * q : xr0 + 64 * (q - TS), xbB = q + 1 < TS ? 64 * (q + 1) : xr0 + 64 * (q + 1 - TS);
                        const bool onA = xbA <= xz && xbA + 63 >= xa && xbA < G, onB = xbB <= xz && xbB + 63 >= xa && xbB < G;      // (wave-uniform)
                        if (onA && onB) FIG_PB_TILE2(q, q + 1, xbA, xbB);
                        else if (onA) FIG_PB_TILE1(q, xbA);
                        else if (onB) FIG_PB_TILE1(q + 1, xbB);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 2 * TS; q++) {
                        const int xb0 = q < TS ? 64 * q : xr0 + 64 * (q - TS);
                        if (!((q % ngrp) == grp && xb0 <= xz && xb0 + 63 >= xa && xb0 < G)) continue;      // (wave-uniform)
                        FIG_PB_TILE1(q, xb0);
                    }
                }
#undef FIG_PB_TILE1
#undef FIG_PB_TILE2
                FIG_PTICK(E, 25);
            }
        }
#endif
        FIG_TICK(E, 29);
        FIG_SYNC();
        FIG_TICK(E, 30);
    }
#ifndef FIG_EMU
    // the accumulators are countsGap of the owned (column, base) pairs: every other gap column stays 0 (placeReads zeroed them)
#pragma unroll
    for (int q = 0; q < 2 * TS; q++) {
        const int xb0 = q < TS ? 64 * q : xr0 + 64 * (q - TS);
        const bool mine = ngrp == 1 || (q % ngrp) == grp;
        const int x = xb0 + lane;
        if (mine && x < G) U.cnt[myb * cg + x] = acc[q];
    }
#endif
    // ---- maxLikelihood += maxProb of every read that had a placement, in file order (:3258-3261)
    if (first_wave) {
        for (int r0 = 0; r0 < nproc; r0 += wsz) {
            const int r = r0 + lane;
            double term = 0.0; int ok = 0;
            if (r < nproc) { term = U.maxlv[r]; ok = U.pflag[r * 2]; }
            const unsigned long long m = fig_ballot(ok != 0);
            const int cnt = nproc - r0 < wsz ? nproc - r0 : wsz;
            for (int k = 0; k < cnt; k++) {
                if (!((m >> k) & 1ULL)) continue;
#ifdef FIG_EMU
                maxLikelihood += term;
#else
                long long bits; memcpy(&bits, &term, 8);
                const unsigned lo32 = (unsigned)__builtin_amdgcn_readlane((int)(bits & 0xffffffffLL), k), hi32 = (unsigned)__builtin_amdgcn_readlane((int)(bits >> 32), k);
                const long long b2 = (long long)(((unsigned long long)hi32 << 32) | lo32);
                double tv; memcpy(&tv, &b2, 8);
                maxLikelihood += tv;
#endif
            }
        }
    }
    FIG_SYNC();                                        // the caller re-initialises pflag for the MLE pass: not before the first wave has read it
    maxLikelihood_out = maxLikelihood; pfl_out += pfl;
}

template <bool LDS>
FIG_D void fig_partial_estep(FigEng &E, int gapoffset, int nproc, double &maxLikelihood, unsigned long long &pfl) {
    const int ts = (E.M->L - 1 + 63) >> 6;
    if (ts <= 2) fig_partial_estep_t<LDS, 2>(E, gapoffset, nproc, maxLikelihood, pfl);
    else fig_partial_estep_t<LDS, 4>(E, gapoffset, nproc, maxLikelihood, pfl);
}

// ---- MLE pass over the partial reads (:3267-3523): arg-max placement per read ("first maximum wins"), accept test,
// pflag / ppos_org records.  One read per wave, two placements per lane, no workgroup barrier.
template <bool LDS>
FIG_NOINLINE FIG_D void fig_partial_mle(FigEng &E, int gapoffset_, int nproc_, unsigned long long &pfl_out, unsigned long long &pml_out) {
    const FigPartU U = fig_part_uniforms(E);
    const fig_plsp SL = (fig_plsp)E.S;
    const int gapoffset = fig_u(gapoffset_), nproc = fig_u(nproc_);
    const long long pb = U.pb;
    const int G = U.G, G0 = U.G0, left = U.left, prc = U.prc;
    const int wave = U.wave, lane = E.lane, nw = U.nw, wsz = U.wsz;
    (void)SL;
    const double *C = fig_c_ptr<LDS>(E);
    const fig_cdp mtf = (fig_cdp)fig_uptr(E.mt_fwd);
    const int ncolE = U.ncolE, xoff = U.xoff;
    unsigned long long pfl = 0;
    for (int p = wave; p < nproc; p += nw) {
        FIG_PT0();
        const int len = U.len[pb + p], flag1 = U.aux[pb + p];
        const long long woff = U.woff[pb + p];
        const bool leftside = flag1 == 1 || flag1 == 4;
        const int j0 = leftside ? 2 : 0, j1 = len - (leftside ? 0 : 2);
        int lo, hi;
        fig_window_partial_u(U, U.pos[pb + p], U.refpos[pb + p], len, gapoffset, lo, hi);
        const fig_cu32p pk = U.packed_c + woff;
        FigBest best; best.v = -FIG_DBL_MAX; best.o = FIG_NOPOS;
        FIG_PTICK(E, 13);
        for (int ob = lo; ob <= hi; ob += 2 * wsz) {
            const int oa = ob + lane, oc = oa + wsz;
            const bool va = oa <= hi, vc = oc <= hi;
            double qa = 1.0, qc = 1.0;
            fig_pchain_m2<LDS>(C, ncolE, pk, mtf, j0, j1, (va ? oa : lo) + xoff, (vc ? oc : lo) + xoff, qa, qc);
            if (va) { if (qa > best.v) { best.v = qa; best.o = oa; } pfl += (unsigned long long)(j1 - j0); }
            if (vc) { if (qc > best.v) { best.v = qc; best.o = oc; } pfl += (unsigned long long)(j1 - j0); }
        }
        FIG_PTICK(E, 14);
        best = fig_wave_best(E, best);
        if (lane == 0 && p < prc) {
            const int o = best.o == FIG_NOPOS ? -left : best.o;
            const double temp_log_val = -fig_log10(best.v);
            if (temp_log_val < U.cutoff) {
                fig_atomic_add_i32(&E.S->valid_count, 1);
                U.pflag[p * 2 + 1] = o;
                if (G == G0) { U.ppos_org[p * 3] = 1; U.ppos_org[p * 3 + 1] = o; U.ppos_org[p * 3 + 2] = len; }
            } else { fig_atomic_add_i32(&E.S->invalid_count, 1); U.pflag[p * 2] = 0; }
        }
        FIG_PTICK(E, 15);
    }
    pfl_out += pfl; pml_out += pfl;
}


// ---- detect_overlap_gapestimate (Figbird.cpp:2513-2779) for placeReads' per-call use (:3501), all threads.  The serial form
// (fig_detect_overlap, lane 0) costs a quarter of a partial-mode pass: one lane walks every read base by base and then every
// (left-hanging, right-hanging) pair of reads.  Here: the reads' byte codes are staged once in the idle table area (LDS),
// l_max / r_min and the spanning-read test run one read per thread, and the pairs are dealt over the threads; the reference's
// "first pair with the largest overlap wins" (strict >, i over the left list, j over the right list, both in read order) is
// the lexicographically smallest (left read, right read) among the pairs of maximal length.  Falls back to the serial form
// when the reads do not fit the staging area.  ret_v is valid in thread 0 (as are S.psr_temp); ends with a barrier.
template <bool LDS>
FIG_D void fig_detect_overlap_par(FigEng &E, const int *pflag, int stride, int gaplen, int *ret_v, int len_thresh) {
    FigState &S = *E.S;
    const int prc = S.partial_read_count, np = E.g->nP;
    const int rl = (E.M->L + 15) & ~15;                 // bytes per staged read: whole packed words (16 bases each)
    const long long cap_bytes = (long long)E.ncolE * 72;
    if (!LDS || !E.pq_lds || E.tiles > 0 || prc > 192 || (long long)prc * rl > cap_bytes || np < prc) {
        if (E.tid == 0) fig_detect_overlap(E, pflag, stride, gaplen, ret_v, len_thresh);
        FIG_SYNC();
        return;
    }
    const FigDevReads &PR = E.B->p;
    const long long pb = E.g->pBase;
    unsigned char *codes = (unsigned char *)(fig_lds + E.off_pq);
    // stage the reads (the MLE table in this area is dead: the next computeProbsGap rebuilds {P,Q})
    FIG_SYNC();
    if (E.tid == 0) { S.ibuf[0] = -FIG_MAX_GAP; S.ibuf[1] = FIG_MAX_GAP; S.ibuf[2] = 0; S.ibuf[3] = 0; S.ibuf[4] = 0; S.ibuf[5] = 0x7fffffff; S.ibuf[6] = 0; }
    // one packed word (16 bases) per thread and trip: two global loads and four LDS stores per 16 codes (a byte at a time through
    // fig_read_code was 5 000 trips of three dependent global loads per call)
    {
        const int nwr = rl >> 4;
        uint32_t *codes32 = (uint32_t *)codes;
        for (int i = E.tid; i < prc * nwr; i += E.nt) {
            const int k = i / nwr, wi = i - k * nwr;
            const int len = PR.len[pb + k];
            const long long woff = PR.woff[pb + k];
            const int nw2 = (len + 15) >> 4;
            uint32_t w = 0, m = 0;
            if (wi < nw2) { w = E.B->packed[woff + wi]; m = (E.B->packed[woff + nw2 + (wi >> 1)] >> ((wi & 1) * 16)) & 0xffffu; }
            for (int q4 = 0; q4 < 4; q4++) {
                uint32_t v = 0;
                for (int q = 0; q < 4; q++) {
                    const int b = q4 * 4 + q, j = wi * 16 + b;
                    const uint32_t c = j < len ? (((m >> b) & 1u) ? 4u : ((w >> (2 * b)) & 3u)) : 255u;
                    v |= c << (8 * q);
                }
                codes32[(long long)k * (rl >> 2) + wi * 4 + q4] = v;
            }
        }
    }
    FIG_SYNC();
    // ---- l_max / r_min (:2545-2590): the reference's per-base walk in closed form
    const int G = S.G, sl = S.side_limit;
    {
        int lm = -FIG_MAX_GAP, rm = FIG_MAX_GAP;
        const int n1 = prc < FIG_READ_CAP + 1 ? prc : FIG_READ_CAP + 1;
        for (int k = E.tid; k < n1; k += E.nt) {
            if (pflag[k * stride] == 0) continue;
            const int match = PR.aux[pb + k], len = PR.len[pb + k], pos = pflag[k * stride + 1];
            // j where the walk stops: the first j with pos + j == gaplen, else len; start: the first position inside [0, gaplen)
            const int jend = (pos <= gaplen && gaplen - pos <= len - 1) ? gaplen - pos : len;
            const int first = pos > 0 ? pos : 0;
            const int start = (first < gaplen && first <= pos + len - 1) ? first : -1;
            if (match == 1 || match == 4 || pos < 0) { if (pos + jend - 1 > lm) lm = pos + jend - 1; }
            else { if (start < rm) rm = start; }
        }
        if (lm != -FIG_MAX_GAP) fig_atomic_max_i32(&S.ibuf[0], lm);
        if (rm != FIG_MAX_GAP) fig_atomic_min_i32(&S.ibuf[1], rm);
    }
    FIG_SYNC();
    int l_max = S.ibuf[0], r_min = S.ibuf[1];
    if (l_max == -FIG_MAX_GAP) l_max = -1;
    if (r_min == FIG_MAX_GAP) r_min = gaplen;
    // ---- spanning reads and the two crossing lists (:2596-2636): kind[k] bit 0 = left list, bit 1 = right list, bit 2 = smflag
    unsigned char *kind = E.scr.smflag;                 // [capP + 8] bytes in the slab
    for (int k = E.tid; k < prc; k += E.nt) {
        int kd = 0;
        if (pflag[k * stride] == 1) {
            const int len = PR.len[pb + k], mr = PR.aux[pb + k];
            const int placed_pos = pflag[k * stride + 1];
            if (placed_pos < 0 && placed_pos + len > G) {
                const unsigned char *rs1 = codes + (long long)k * rl, *l, *r;
                int ln = (-placed_pos < sl) ? fig_substr(rs1, len, 0, -placed_pos, &l) : fig_substr(rs1, len, -placed_pos - sl, sl, &l);
                int rn = fig_substr(rs1, len, -placed_pos + G, sl, &r);
                if (fig_find_partial_match(S.gap_left, S.gl_len, l, ln, 0, 0, len_thresh) &&
                    fig_find_partial_match(S.gap_right, S.gr_len, r, rn, 1, 0, len_thresh)) fig_atomic_or_i32(&S.ibuf[2], 1);
            }
            if (placed_pos < 0 && placed_pos + len - 1 >= r_min && placed_pos + len <= G) kd |= 1;
            if (placed_pos > 0 && placed_pos <= l_max) kd |= 2;
            if (placed_pos < 0 && placed_pos + len > G && (mr == 2 || mr == 3)) kd |= 2 | 4;
        }
        kind[k] = (unsigned char)kd;
    }
    FIG_SYNC();
    const int ovflag = S.ibuf[2];
    if (ovflag || (E.g->stat2 == 1 && E.g->G0 <= 20 && gaplen == E.g->stat3)) {
        for (int k = E.tid; k < prc; k += E.nt) E.scr.smflag[k] = (kind[k] >> 2) & 1;      // as the serial form leaves it (every thread: the entries it wrote)
        if (E.tid == 0) { ret_v[0] = 300; ret_v[1] = 0; }
        FIG_SYNC();
        return;
    }
    if (r_min <= l_max) {
        // ---- the pairs (:2650-2757)
        int best_len = 0, best_key = 0x7fffffff, bad = 0;
        const int ot = S.overlap_threshold;
        for (int pi = E.tid; pi < prc * prc; pi += E.nt) {
            const int li = pi / prc, rj = pi - li * prc;
            if (li == rj || !(kind[li] & 1) || !(kind[rj] & 2)) continue;
            const int len = PR.len[pb + li], len2 = PR.len[pb + rj];
            const int p1 = pflag[li * stride + 1], p2 = pflag[rj * stride + 1];
            int diff_gap = p1 + len - gaplen, overlap_count;
            if (diff_gap > 0) overlap_count = (p1 + len - 1) - p2 + 1 - diff_gap;
            else { overlap_count = (p1 + len - 1) - p2 + 1; diff_gap = 0; }
            if (overlap_count < ot) continue;
            const unsigned char *rs1 = codes + (long long)li * rl, *rs2 = codes + (long long)rj * rl;
            const unsigned char *cl = rs1, *cr = rs2, *fr;
            int len1 = 0, lenr = 0, fn;
            if (!(kind[rj] & 4)) {
                fn = fig_get_read_frag(E, rs1, len, p1, &fr);
                if (fig_find_partial_match(S.gap_left, S.gl_len, fr, fn, 0, 1, -1)) len1 = fig_substr(rs1, len, len - overlap_count - diff_gap, overlap_count, &cl);
                fn = fig_get_read_frag(E, rs2, len2, p2, &fr);
                if (fig_find_partial_match(S.gap_right, S.gr_len, fr, fn, 1, 1, -1)) lenr = fig_substr(rs2, len2, 0, overlap_count, &cr);
            } else {
                const int x = p2;
                fn = fig_get_read_frag(E, rs1, len, p1, &fr);
                if (fig_find_partial_match(S.gap_left, S.gl_len, fr, fn, 0, 1, -1)) len1 = fig_substr(rs1, len, len - overlap_count - x, overlap_count - x, &cl);
                fn = fig_substr(rs2, len2, -x + G, sl, &fr);
                if (fig_find_partial_match(S.gap_right, S.gr_len, fr, fn, 1, 1, -1)) lenr = fig_substr(rs2, len2, -x, overlap_count + x, &cr);
            }
            if (len1 > 0 && lenr > 0 && len1 == lenr) {
                if (fig_get_diff(cl, cr, len1) <= .1) { if (len1 > best_len || (len1 == best_len && pi < best_key)) { best_len = len1; best_key = pi; } }
                else bad = 1;
            }
        }
        if (best_len > 0) fig_atomic_max_i32(&S.ibuf[4], best_len);
        if (bad) fig_atomic_or_i32(&S.ibuf[6], 1);
        FIG_SYNC();
        const int max_overlap = S.ibuf[4];
        if (best_len > 0 && best_len == max_overlap) fig_atomic_min_i32(&S.ibuf[5], best_key);
        FIG_SYNC();
        for (int k = E.tid; k < prc; k += E.nt) E.scr.smflag[k] = (kind[k] >> 2) & 1;
        if (E.tid == 0) {
            const int false_overlap_flag = S.ibuf[6] ? -1 : 0;
            if (max_overlap > 0) { const int key = S.ibuf[5]; S.psr_temp[0] = key / prc; S.psr_temp[1] = key - (key / prc) * prc; }
            if ((false_overlap_flag == 0 && max_overlap >= ot) || (false_overlap_flag == -1 && max_overlap >= 2 * ot)) { ret_v[0] = max_overlap; ret_v[1] = 0; }
            else if (false_overlap_flag == -1 || max_overlap < ot) { ret_v[0] = 0; ret_v[1] = -1; S.psr_temp[0] = S.psr_temp[1] = -1; }
        }
        FIG_SYNC();
        return;
    }
    for (int k = E.tid; k < prc; k += E.nt) E.scr.smflag[k] = (kind[k] >> 2) & 1;
    if (E.tid == 0) { ret_v[0] = 0; ret_v[1] = 0; }
    FIG_SYNC();
}

#endif
