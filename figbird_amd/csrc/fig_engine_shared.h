// fig_engine_shared.h -- the shared-factor form of the unmapped E-step's placement phase (gfx950 only).
//
// The per-base factor of placeReads' E-step product (Figbird.cpp:3559-3589)
//        f = probs[x][b] * (1 - e[k]) + e[k] * errorProbs[x][b],        x = o + j,  b = read base j,  k = j or L-1-j
// does not depend on the READ, only on (column x, chain step j, orientation, base): every read of the gap that puts base b at
// step j of placement o multiplies by the same double.  The pair-chain form (fig_hot_estep) recomputes it per (read,
// placement, step): 4 FP64 operations and 16 B of LDS per step.  Here a lane owns ONE placement o and carries the products
// of a whole CHUNK of 32 reads; per chain step it fetches the column's {P,Q}[0..3] once (64 B), computes the eight factors
// f[orientation][base] once (24 operations: the same multiply, multiply, add as the reference, so the same bits) and every
// read's product takes the factor its own base selects with ONE multiply:  (24 + 32) / 32 = 1.75 operations and 2 B of LDS
// per (read, placement, step) instead of 4 and 16.
//
// The select is wave-uniform (all lanes of a wave work on the same reads) but differs per read and step, so it is a
// register-file index, not a branch: the eight factors sit in v[232:247], the constant 1.0 in v[248:249], the wave runs in
// VGPR-index mode (s_set_gpr_idx_on ... SRC0) and per multiply ONE SALU instruction moves the next 16-bit entry of the
// gap's operand-select stream (fig_stream_kernel: 0x1000 | 2 * (4 * reverse + base)) into M0.  One SALU instruction per
// multiply is what the CU's scalar unit sustains beside the FP64 pipe (a 4-bit stream that needed two was scalar-bound);
// SALU and VALU issue from the two waves of a SIMD side by side; measured (tools/ubench/idx2.hip) the indexed multiplies
// cost what plain ones do.
//
// Work split of a chunk: tiles of 64 consecutive placements; wave w takes tiles w, w + nw, ... with all 32 reads; the tiles
// left over by the last full round are split over the waves by READS (32/f reads each) so that no wave idles.  The finished
// weights stay in registers (two banks of 32 doubles per lane) and go to the LDS weight rows four reads at a time, where
// the unchanged column pass adds them in the reference's (read, placement) order.  Reads the stream cannot describe (an N
// base, a length other than L) take the generic chain of fig_engine_hot.h inside their row group.
#ifndef FIG_ENGINE_SHARED_H
#define FIG_ENGINE_SHARED_H
#ifndef FIG_EMU
#define FIG_FI __device__ static __forceinline__

// ---- wave maximum of a non-negative double through its bit pattern (positive doubles order like their 64-bit integers):
// v_max_u32 with DPP row shifts / broadcasts on the high words, then on the low words of the lanes that hold the maximal
// high word.  Result in every lane (hi, lo).
FIG_D unsigned fig_wave_max_u32(unsigned v) {
#define FIG_DPP_UMAX(ctrl, rmask) do { const unsigned y_ = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xf, false); v = y_ > v ? y_ : v; } while (0)
    FIG_DPP_UMAX(0x111, 0xf); FIG_DPP_UMAX(0x112, 0xf); FIG_DPP_UMAX(0x114, 0xf); FIG_DPP_UMAX(0x118, 0xf);
    FIG_DPP_UMAX(0x142, 0xa); FIG_DPP_UMAX(0x143, 0xc);
#undef FIG_DPP_UMAX
    return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// ---- log10 for the E-step weights (`fig_log10_fast` in DESIGN.md), ~40 FP64 operations against ~105 of the library's (which became a quarter of the E-step's
// vector work once the chains cost 1.75 operations a step): x = m 2^e with m in [sqrt(1/2), sqrt(2)), s = (m-1)/(m+1) through
// v_rcp_f64 + one Newton step with the quotient's rounding error recovered (s_lo), log m = 2 s + s z q(z) (atanh series in
// z = s^2 to z^11), log10 x = e log10(2) + log m log10(e) assembled from hi/lo parts.  Against 80-bit log10l on 3e7 arguments
// over (0, 1] incl. denormals and the neighbourhood of 1 (tools/ubench/fast_log10_check.c: the same operations on the CPU):
// max error 1.87 ulp, > 1 ulp in 1.2e-5 of the arguments -- glibc's log10, which the reference calls, is > 1 ulp in 2.3e-4 of
// them (max 1.58); the two agree bit for bit on 99.44 %.  Exact for 0 (-inf) and 1 (0).  Used for the weights only; the
// per-read maximum (maxlv) keeps the library's log10.
// (This is the log10 stage of fig_weights_n below; there is no scalar copy of it to keep in step.)

// ---- w = exp(0.5 log10 p) for N values at once, every stage applied to all N before the next one: N independent dependent
// chains side by side, which is what fills the FP64 pipe at two waves per SIMD (the compiler keeps one evaluation's ~75
// operations together if they are written one evaluation after the other), and each 64-bit constant is materialised once per
// stage instead of once per evaluation.  log10 as described above; exp(x), x in
// [-162, 0] or -inf: k = rint(x log2 e), r = x - k ln 2 as an exact high part and a small low part, exp(r) = 1 + r + r^2 q(r)
// with the Taylor polynomial to r^13 (|r| <= 0.347: truncation 4e-18) and 1 + r_hi summed exactly, scaled by 2^k: 29 FP64
// operations against the library's 43, max 0.69 ulp, equal to glibc's exp bit for bit on 98.4 % of the range (checked on the
// host with the same operation sequence; the weights are compared at 1e-6).
template <int N>
FIG_FI void fig_weights_n(const double (&x)[N], double (&w)[N]) {
    double m[N], a[N], b[N], r[N], s[N], z[N], q[N], s_lo[N], t[N], lm_lo[N], ed[N], p_hi[N], p_lo[N], r_hi[N], sum[N], err[N], tl[N];
    int e[N];
#pragma unroll
    for (int i = 0; i < N; i++) { m[i] = __builtin_amdgcn_frexp_mant(x[i]); e[i] = __builtin_amdgcn_frexp_exp(x[i]); }
#pragma unroll
    for (int i = 0; i < N; i++) { const bool lowhalf = m[i] < 0.70710678118654752440; m[i] = lowhalf ? m[i] + m[i] : m[i]; e[i] = lowhalf ? e[i] - 1 : e[i]; }
#pragma unroll
    for (int i = 0; i < N; i++) { a[i] = m[i] - 1.0; b[i] = m[i] + 1.0; }
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = __builtin_amdgcn_rcp(b[i]);
#pragma unroll
    for (int i = 0; i < N; i++) { const double e0 = __builtin_fma(-b[i], r[i], 1.0); r[i] = __builtin_fma(r[i], e0, r[i]); }
#pragma unroll
    for (int i = 0; i < N; i++) { s[i] = a[i] * r[i]; z[i] = s[i] * s[i]; q[i] = 2.0 / 23.0; }
#define FIG_W_H(c) _Pragma("unroll") for (int i = 0; i < N; i++) q[i] = __builtin_fma(q[i], z[i], c);
    FIG_W_H(2.0 / 21.0) FIG_W_H(2.0 / 19.0) FIG_W_H(2.0 / 17.0) FIG_W_H(2.0 / 15.0) FIG_W_H(2.0 / 13.0)
    FIG_W_H(2.0 / 11.0) FIG_W_H(2.0 / 9.0) FIG_W_H(2.0 / 7.0) FIG_W_H(2.0 / 5.0) FIG_W_H(2.0 / 3.0)
#undef FIG_W_H
    const double L2hi = 0x1.34413509f78p-2, L2lo = 0x1.fef311f12b358p-46;      // log10(2): hi holds 41 bits, e * L2hi is exact
    const double IE_hi = 0x1.bcb7b1526e50ep-2, IE_lo = 0x1.95355baaafad3p-57;    // log10(e)
#pragma unroll
    for (int i = 0; i < N; i++) { s_lo[i] = __builtin_fma(-s[i], b[i], a[i]) * r[i]; t[i] = s[i] * z[i] * q[i]; }
#pragma unroll
    for (int i = 0; i < N; i++) { const double lm_hi = 2.0 * s[i]; lm_lo[i] = __builtin_fma(2.0, s_lo[i], t[i]); ed[i] = (double)e[i]; p_hi[i] = lm_hi * IE_hi;
                                  p_lo[i] = __builtin_fma(lm_hi, IE_hi, -p_hi[i]) + __builtin_fma(lm_hi, IE_lo, lm_lo[i] * IE_hi); r_hi[i] = ed[i] * L2hi; }
#pragma unroll
    for (int i = 0; i < N; i++) { sum[i] = r_hi[i] + p_hi[i]; err[i] = (r_hi[i] - sum[i]) + p_hi[i]; if (e[i] == 0) { sum[i] = p_hi[i]; err[i] = 0.0; } }
#pragma unroll
    for (int i = 0; i < N; i++) { const double res = sum[i] + (err[i] + __builtin_fma(ed[i], L2lo, p_lo[i])); tl[i] = x[i] == 0.0 ? -__builtin_inf() : res; }
    // exp(0.5 tl)
    const double L2E = 0x1.71547652b82fep+0, LN2_HI = 0x1.62e42fee00000p-1, LN2_LO = 0x1.a39ef35793c76p-33;
    double y[N], kd[N], rh[N], rl[N], rr[N], g[N];
#pragma unroll
    for (int i = 0; i < N; i++) { y[i] = 0.5 * tl[i]; kd[i] = __builtin_rint(y[i] * L2E); }
#pragma unroll
    for (int i = 0; i < N; i++) { rh[i] = __builtin_fma(-kd[i], LN2_HI, y[i]); rl[i] = -kd[i] * LN2_LO; rr[i] = rh[i] + rl[i]; g[i] = 1.0 / 6227020800.0; }
#define FIG_W_H(c) _Pragma("unroll") for (int i = 0; i < N; i++) g[i] = __builtin_fma(g[i], rr[i], c);
    FIG_W_H(1.0 / 479001600.0) FIG_W_H(1.0 / 39916800.0) FIG_W_H(1.0 / 3628800.0) FIG_W_H(1.0 / 362880.0) FIG_W_H(1.0 / 40320.0) FIG_W_H(1.0 / 5040.0)
    FIG_W_H(1.0 / 720.0) FIG_W_H(1.0 / 120.0) FIG_W_H(1.0 / 24.0) FIG_W_H(1.0 / 6.0) FIG_W_H(0.5)
#undef FIG_W_H
#pragma unroll
    for (int i = 0; i < N; i++) {
        const double tt = rr[i] * rr[i] * g[i];
        const double ss = 1.0 + rh[i], ee = (1.0 - ss) + rh[i];
        const double res = __builtin_ldexp(ss + ((ee + rl[i]) + tt), (int)kd[i]);
        w[i] = y[i] < -745.0 ? 0.0 : res;
    }
}

// FigState lives in LDS; through E.S (a generic pointer) its fields are FLAT loads too: the staging arrays go through this view
typedef FigState __attribute__((address_space(3))) *fig_lsp;

// ---- the indexed multiplies.  F0..F7 and the constant 1.0 are pinned to v[232:249]; p* are the products (any registers);
// w* hold two 16-bit entries of the operand-select stream each: the value M0 takes (0x1000 = "SRC0 is indexed" | register
// offset 2 * (4 * reverse + base), or 16 = the 1.0 for a slot without a regular read).  Per multiply ONE SALU instruction.
// The blocks write M0 (and leave index mode off again).  M0 cannot be named in the clobber list: the AMDGPU backend treats it as a
// reserved register ("inline asm clobber list contains reserved registers: m0 ... may not be preserved"), i.e. the list entry
// is ignored and only buys a warning per instantiation.  What makes the blocks safe is that nothing else in these kernels
// lives in M0 across them: gfx950 needs M0 for s_sendmsg, LDS-DMA, s_movrel and the GWS/ordered-count DS operations only,
// none of which the engine uses -- tools/isa_m0_check.py verifies on the compiled ISA that every M0 access of a function that
// enters index mode belongs to one of these blocks.
#define FIG_SH_M2(pa, pb, w) \
    "s_and_b32 m0, " w ", 0xffff\n v_mul_f64 " pa ", v[232:233], " pa "\n" \
    "s_lshr_b32 m0, " w ", 16\n v_mul_f64 " pb ", v[232:233], " pb "\n"
#define FIG_SH_FIN "{v[232:233]}"(f[0]), "{v[234:235]}"(f[1]), "{v[236:237]}"(f[2]), "{v[238:239]}"(f[3]), "{v[240:241]}"(f[4]), \
                   "{v[242:243]}"(f[5]), "{v[244:245]}"(f[6]), "{v[246:247]}"(f[7]), "{v[248:249]}"(one)
FIG_FI void fig_sh_mul4(double *p, const double *f, double one, const uint32_t *w) {
    asm volatile("s_set_gpr_idx_on %[w0], 1\n" FIG_SH_M2("%[p0]", "%[p1]", "%[w0]") FIG_SH_M2("%[p2]", "%[p3]", "%[w1]") "s_set_gpr_idx_off\n"
                 : [p0] "+v"(p[0]), [p1] "+v"(p[1]), [p2] "+v"(p[2]), [p3] "+v"(p[3])
                 : FIG_SH_FIN, [w0] "s"(w[0]), [w1] "s"(w[1]) : "scc");
}
FIG_FI void fig_sh_mul8(double *p, const double *f, double one, const uint32_t *w) {
    asm volatile("s_set_gpr_idx_on %[w0], 1\n" FIG_SH_M2("%[p0]", "%[p1]", "%[w0]") FIG_SH_M2("%[p2]", "%[p3]", "%[w1]")
                 FIG_SH_M2("%[p4]", "%[p5]", "%[w2]") FIG_SH_M2("%[p6]", "%[p7]", "%[w3]") "s_set_gpr_idx_off\n"
                 : [p0] "+v"(p[0]), [p1] "+v"(p[1]), [p2] "+v"(p[2]), [p3] "+v"(p[3]), [p4] "+v"(p[4]), [p5] "+v"(p[5]), [p6] "+v"(p[6]), [p7] "+v"(p[7])
                 : FIG_SH_FIN, [w0] "s"(w[0]), [w1] "s"(w[1]), [w2] "s"(w[2]), [w3] "s"(w[3]) : "scc");
}
FIG_FI void fig_sh_mul16(double *p, const double *f, double one, const uint32_t *w) {
    asm volatile("s_set_gpr_idx_on %[w0], 1\n" FIG_SH_M2("%[p0]", "%[p1]", "%[w0]") FIG_SH_M2("%[p2]", "%[p3]", "%[w1]")
                 FIG_SH_M2("%[p4]", "%[p5]", "%[w2]") FIG_SH_M2("%[p6]", "%[p7]", "%[w3]") FIG_SH_M2("%[p8]", "%[p9]", "%[w4]")
                 FIG_SH_M2("%[p10]", "%[p11]", "%[w5]") FIG_SH_M2("%[p12]", "%[p13]", "%[w6]") FIG_SH_M2("%[p14]", "%[p15]", "%[w7]") "s_set_gpr_idx_off\n"
                 : [p0] "+v"(p[0]), [p1] "+v"(p[1]), [p2] "+v"(p[2]), [p3] "+v"(p[3]), [p4] "+v"(p[4]), [p5] "+v"(p[5]), [p6] "+v"(p[6]), [p7] "+v"(p[7]),
                   [p8] "+v"(p[8]), [p9] "+v"(p[9]), [p10] "+v"(p[10]), [p11] "+v"(p[11]), [p12] "+v"(p[12]), [p13] "+v"(p[13]), [p14] "+v"(p[14]), [p15] "+v"(p[15])
                 : FIG_SH_FIN, [w0] "s"(w[0]), [w1] "s"(w[1]), [w2] "s"(w[2]), [w3] "s"(w[3]), [w4] "s"(w[4]), [w5] "s"(w[5]), [w6] "s"(w[6]), [w7] "s"(w[7])
                 : "scc");
}

// ---- the chain: NS products per lane over the L steps of the chunk.  colp = PQ + x (x = o + xoff: the lane's first column),
// st = the chunk's stream row of step 0 at this wave's first slot (16 dwords per step; one padded row behind the last).
// The loads of step j + 1 (four ds_read_b128, the stream row, the {1-e, e} pairs of both orientations) are issued before
// the arithmetic of step j.  (A scalar load that misses the scalar cache comes back from L2 in ~190 cycles against ~135 on
// a hit, tools/ubench/kcache.hip: one step of arithmetic covers either.)
// One chain step: issue the loads of step jn (four ds_read_b128, the stream row, the {1-e, e} pairs of both orientations)
// into the `n` set, then the eight factors and NS indexed multiplies of the current step from the `c` set.
template <int NS>
FIG_FI void fig_sh_step(double (&p)[FIG_SH_C], const uint32_t (&wc)[NS / 2], const double (&kc)[4], const FigPQ (&ac)[4],
                        uint32_t (&wn)[NS / 2], double (&kn)[4], FigPQ (&an)[4], const FigPQ *colp, int ncolE, fig_cu32p st, fig_cdp ktf, fig_cdp ktr, int jn) {
    {
        // The current step's scalar loads (stream row, {1-e, e} pairs) had a whole step to arrive: wait for them BEFORE the next
        // step's loads go out.  Scalar loads return out of order, so the only wait that covers them is lgkmcnt(0), and placed
        // at their first use (where the compiler puts it) it would also drain the four ds_read_b128 just issued for the next step.
        __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0), vmcnt / expcnt untouched
        fig_cu32p sn = st + jn * (FIG_SH_C / 2);
#pragma unroll
        for (int k = 0; k < NS / 2; k++) wn[k] = sn[k];
        kn[0] = ktf[2 * jn]; kn[1] = ktf[2 * jn + 1]; kn[2] = ktr[2 * jn]; kn[3] = ktr[2 * jn + 1];
        const FigPQ *cn = colp + jn;
#pragma unroll
        for (int b = 0; b < 4; b++) an[b] = cn[b * ncolE];
    }
    // f[b] = P (1 - e[j]) + e[j] Q for forward reads (:3581-3589), f[4 + b] with {1 - e[L-1-j], e[L-1-j]} for reverse reads
    // (:3569-3576); the sixteen products first and the eight sums after them, so that no sum waits for its own products
    double f[8], m1[8], m2[8];
#pragma unroll
    for (int b = 0; b < 4; b++) { m1[b] = ac[b].p * kc[0]; m1[4 + b] = ac[b].p * kc[2]; }
#pragma unroll
    for (int b = 0; b < 4; b++) { m2[b] = kc[1] * ac[b].q; m2[4 + b] = kc[3] * ac[b].q; }
#pragma unroll
    for (int i = 0; i < 8; i++) f[i] = m1[i] + m2[i];
    const double one = 1.0;
    if (NS == 32) { fig_sh_mul16(&p[0], f, one, &wc[0]); fig_sh_mul16(&p[16], f, one, &wc[8]); }
    else if (NS == 16) fig_sh_mul16(&p[0], f, one, &wc[0]);
    else if (NS == 8) fig_sh_mul8(&p[0], f, one, &wc[0]);
    else fig_sh_mul4(&p[0], f, one, &wc[0]);
}

// ---- the chain: NS products per lane over the L steps of the chunk.  colp = PQ + x (x = o + xoff: the lane's first column),
// st = the chunk's stream row of step 0 at this wave's first slot (16 dwords per step; one padded row behind the last).
// The loads of step j + 1 are issued before the arithmetic of step j; two register sets take turns (no copies).  (A scalar
// load that misses the scalar cache comes back from L2 in ~190 cycles against ~135 on a hit, tools/ubench/kcache.hip: one
// step of arithmetic covers either.)
template <int NS>
FIG_FI void fig_sh_chain(double (&p)[FIG_SH_C], const FigPQ *colp, int ncolE, fig_cu32p st, fig_cdp ktf, fig_cdp ktr, int L) {
    uint32_t w0[NS / 2], w1[NS / 2];
    double k0[4], k1[4];
    FigPQ a0[4], a1[4];
#pragma unroll
    for (int k = 0; k < NS / 2; k++) w0[k] = st[k];
    k0[0] = ktf[0]; k0[1] = ktf[1]; k0[2] = ktr[0]; k0[3] = ktr[1];
#pragma unroll
    for (int b = 0; b < 4; b++) a0[b] = colp[b * ncolE];
    int j = 0;
#ifndef FIG_SH_UNROLL
#define FIG_SH_UNROLL 2          // step pairs per loop trip
#endif
    for (; j + 2 * FIG_SH_UNROLL <= L; j += 2 * FIG_SH_UNROLL) {
#pragma unroll
        for (int u = 0; u < FIG_SH_UNROLL; u++) {
            fig_sh_step<NS>(p, w0, k0, a0, w1, k1, a1, colp, ncolE, st, ktf, ktr, j + 2 * u + 1);
            fig_sh_step<NS>(p, w1, k1, a1, w0, k0, a0, colp, ncolE, st, ktf, ktr, j + 2 * u + 2);
        }
    }
    for (; j + 2 <= L; j += 2) {
        fig_sh_step<NS>(p, w0, k0, a0, w1, k1, a1, colp, ncolE, st, ktf, ktr, j + 1);
        fig_sh_step<NS>(p, w1, k1, a1, w0, k0, a0, colp, ncolE, st, ktf, ktr, j + 2);
    }
    if (j < L) fig_sh_step<NS>(p, w0, k0, a0, w1, k1, a1, colp, ncolE, st, ktf, ktr, j + 1);
}

// One unit = (tile of 64 placements, NS reads [s0, s0 + NS) of the chunk): products start from the insert-size terms
// (1.0 for a placement outside the read's window: never used), run the chain and go to the workgroup's product rows in the
// scratch slab, prow[s * pst + (o + L - 1)] (coalesced 512-byte stores; read back four reads at a time by phase B).
// Lanes past the last placement (vo false) run on the tile's first placement and store nothing.
template <int NS>
FIG_FI void fig_sh_unit(const int s0, FigState &S, const FigHotU &U, const FigPQ *PQ, fig_cu32p stream, fig_gdp prow, const int pst, const int o, const bool vo, const int ocalc) {
    // s0 = first read of the unit, counted from the super-chunk's first read (S.sh_* / prow row index); `stream` = its chunk's rows
    const fig_lsp SL = (fig_lsp)&S;
    double p[FIG_SH_C];
#pragma unroll
    for (int i = 0; i < NS; i++) {
        const int s = s0 + i;
        const int lo = fig_u(SL->sh_lo[s]), hi = fig_u(SL->sh_hi[s]), tis0 = fig_u(SL->sh_tis0[s]), dir = (fig_u(SL->sh_aux[s]) & 4) ? 1 : -1;
        p[i] = 1.0;
        if (ocalc >= lo && ocalc <= hi) p[i] = ((fig_gcdp)U.insd)[tis0 + dir * ocalc];
    }
    fig_sh_chain<NS>(p, PQ + (ocalc + U.xoff), U.ncolE, stream + ((s0 & (FIG_SH_C - 1)) >> 1), (fig_cdp)U.kt_fwd, (fig_cdp)U.kt_rev, U.L);
    if (vo) {
        fig_gdp dst = prow + (long long)s0 * pst + (o + U.L - 1);
#pragma unroll
        for (int i = 0; i < NS; i++) dst[(long long)i * pst] = p[i];
    }
}

// ---- the E-step with the shared-factor placement phase.  Same contract as fig_hot_estep<LDS = true, CPL, TILED = false>.
template <int CPL>
FIG_NOINLINE FIG_D void fig_hot_estep_sh(FigEng &E, int gapoffset) {
    constexpr bool LDS = true;
    constexpr int NR = 4;                                      // weight rows per group of the column pass (the class has >= 4)
    constexpr int KP = 2;                                      // placements per thread and row in phase B: Wn <= KP * nt (dispatcher)
    FigState &S = *E.S;
    const fig_lsp SL = (fig_lsp)E.S;
    const FigHotU U = fig_hot_uniforms(E);
    const int G = U.G, nU = U.nU, cg = U.cg;
    const long long ub = U.ub;
    const FigPQ *PQ = fig_pq_ptr<LDS>(E);
    double *W = fig_w_ptr<LDS>(E);
    const int Wcap = U.Wcap;
    const int wave = fig_u(E.wave), lane = E.lane, tid = E.tid, nw = U.nw;
    const fig_cu32p stream0 = (fig_cu32p)(fig_uptr(fig_uptr(E.B)->ustream) + fig_u64(fig_uptr(E.g)->streamOff));
    const long long wbase = nU > 0 ? fig_u64(U.u_woff[ub]) : 0;     // packed-read offsets are kept relative to the gap's first read
    const fig_gdp prow = (fig_gdp)fig_uptr(E.scr.wg);                        // raw products of the chunk: [32 reads][pst] (scratch slab, L2)
    const int pst = Wcap;
    double acc[CPL];
#pragma unroll
    for (int m = 0; m < CPL; m++) acc[m] = 0;
    const int pb_c = wave & 3, pb_x0 = (wave >> 2) * CPL * U.wsz + lane;
    unsigned long long fl_acc = 0;
    uint32_t *plb = (uint32_t *)(fig_lds + fig_u(E.off_plb));
    // tiles of 64 placements over the range [-(L-1), G-1]
    const int Wn = G + U.L - 1;
    const int nT = (Wn + 63) >> 6;
    // Chunks per super-chunk (at most E.sh_on <= FIG_SH_SC): more chunks deal the (chunk, tile) items more evenly over the waves
    int scn = fig_u(E.sh_on);
    scn = scn < 1 ? 1 : (scn > FIG_SH_SC ? FIG_SH_SC : scn);
    FIG_SYNC();                                                // placeReads zeroed countsGap already (:3050-3056)
    FIG_T0(E);
    for (int c0 = 0; c0 < nU; c0 += scn * FIG_SH_C) {
        // ---- a super-chunk: up to scn chunks of 32 reads.  Insert-size windows and packed-read offsets of its reads
        const int nrd = nU - c0 < scn * FIG_SH_C ? nU - c0 : scn * FIG_SH_C;
        const int nc = (nrd + FIG_SH_C - 1) / FIG_SH_C;
        if (tid < FIG_SH_SC * FIG_SH_C) {
            const int r = c0 + tid;
            int lo = 0, hi = -1, tis0 = 0, dir = 1; bool irr = false;
            int len = 0, aux = 0; long long woff = 0;
            if (tid < nrd) {
                len = U.u_len[ub + r]; aux = U.u_aux[ub + r]; woff = U.u_woff[ub + r];
                const FigWin w = fig_window_u(U, U.u_pos[ub + r], len, gapoffset);
                lo = w.lo; hi = w.hi; tis0 = w.tis0; dir = w.dir;
                irr = ((aux >> 1) & 1) != 0 || len != U.L;
            }
            SL->sh_lo[tid] = (short)lo; SL->sh_hi[tid] = (short)hi; SL->sh_tis0[tid] = tis0; SL->sh_len[tid] = (unsigned char)len;
            SL->sh_aux[tid] = (unsigned char)((aux & 3) | (dir > 0 ? 4 : 0)); SL->sh_woff[tid] = (unsigned)(woff - wbase);
            const unsigned long long im = fig_ballot(irr);
            if (lane == 0) { SL->sh_irr[2 * wave] = (unsigned)(im & 0xffffffffULL); SL->sh_irr[2 * wave + 1] = (unsigned)(im >> 32); }
        }
        FIG_SYNC();
        // position lists of the first row group: fetched now, parked in plb when the group starts (wave t: row t)
        uint32_t plv_next = 0;
        if (wave < NR && c0 + wave < nU) {
            const int len = fig_u(SL->sh_len[wave]);
            const int ndw = 2 + ((len + 3) >> 2) + 4;
            if (lane < ndw) plv_next = ((fig_gcu32p)U.packed)[wbase + fig_u((int)SL->sh_woff[wave]) + ((len + 15) >> 4) + ((len + 31) >> 5) + lane];
        }
        FIG_TICK(E, 34);
        // ---- phase A: lanes = placements, the raw products of all reads -> prow.  Work items = (chunk, tile) pairs: full
        // rounds of nw items with all 32 reads of their chunk; the items left over are split by reads over f waves each.
        {
            const int nit = nT * nc;
            const int nfull = nit / nw, mleft = nit - nfull * nw;
            const fig_cu32p stc = stream0 + (long long)(c0 / FIG_SH_C) * U.L * (FIG_SH_C / 2);
            for (int k = 0; k < nfull; k++) {
                const int it = k * nw + wave, ci = it / nT, tl = it - ci * nT;
                const int ob = -(U.L - 1) + 64 * tl, o = ob + lane;
                const bool vo = o <= G - 1;
                fig_sh_unit<32>(ci * FIG_SH_C, S, U, PQ, stc + (long long)ci * U.L * (FIG_SH_C / 2), prow, pst, o, vo, vo ? o : ob);
            }
            if (mleft > 0) {
                int fsplit = 1;
                while (fsplit * 2 * mleft <= nw && fsplit < 8) fsplit *= 2;
                if (wave < mleft * fsplit) {
                    const int nsl = FIG_SH_C / fsplit;                 // reads per wave in the split round
                    const int it = nfull * nw + wave / fsplit, ci = it / nT, tl = it - ci * nT;
                    const int s0 = ci * FIG_SH_C + (wave % fsplit) * nsl;
                    const int ob = -(U.L - 1) + 64 * tl, o = ob + lane;
                    const bool vo = o <= G - 1;
                    const fig_cu32p stp = stc + (long long)ci * U.L * (FIG_SH_C / 2);
                    if (fsplit == 1) fig_sh_unit<32>(s0, S, U, PQ, stp, prow, pst, o, vo, vo ? o : ob);
                    else if (fsplit == 2) fig_sh_unit<16>(s0, S, U, PQ, stp, prow, pst, o, vo, vo ? o : ob);
                    else if (fsplit == 4) fig_sh_unit<8>(s0, S, U, PQ, stp, prow, pst, o, vo, vo ? o : ob);
                    else fig_sh_unit<4>(s0, S, U, PQ, stp, prow, pst, o, vo, vo ? o : ob);
                }
            }
        }
        FIG_TICK(E, 35);
        FIG_SYNC();                                            // every wave's products are in prow
        // ---- phase B: four reads at a time through the LDS weight rows.  Thread tid takes placements tid, tid + nt, ... of
        // every row; the products of the next group are fetched while this group's column pass runs.
        double pn[NR][KP];
#pragma unroll
        for (int t = 0; t < NR; t++)
#pragma unroll
            for (int k = 0; k < KP; k++) { const int i = tid + k * U.nt; pn[t][k] = (i < Wn && c0 + t < nU) ? prow[(long long)t * pst + i] : 0.0; }
        for (int q0 = 0; q0 < nrd; q0 += NR) {
            const unsigned irr = (unsigned)fig_u((int)SL->sh_irr[q0 >> 5]) >> (q0 & 31);      // bits 0..NR-1: the group's reads
            // the group's scalars, all four rows at once (lane t of every wave reads row t's, v_readlane hands them round)
            int g_lo = 0, g_hi = -1;
            if (lane < NR && c0 + q0 + lane < nU) { g_lo = SL->sh_lo[q0 + lane]; g_hi = SL->sh_hi[q0 + lane]; }
            if (wave < NR) {
                plb[wave * 64 + lane] = plv_next;              // fetched while the previous group ran
                plv_next = 0;
                const int sn = q0 + NR + wave;                 // the same row of the next group of this chunk
                if (sn < nrd) {
                    const int len = fig_u(SL->sh_len[sn]);
                    const int ndw = 2 + ((len + 3) >> 2) + 4;
                    if (lane < ndw) plv_next = ((fig_gcu32p)U.packed)[wbase + fig_u((int)SL->sh_woff[sn]) + ((len + 15) >> 4) + ((len + 31) >> 5) + lane];
                }
            }
            double pc[NR][KP];
#pragma unroll
            for (int t = 0; t < NR; t++)
#pragma unroll
                for (int k = 0; k < KP; k++) pc[t][k] = pn[t][k];
            FIG_TICK(E, 19);
            // w = exp(0.5 log10 p) of the group's four rows side by side: four independent dependent chains in one basic block,
            // which the scheduler interleaves (a row's own ~90-operation chain alone leaves the FP64 pipe half empty at two
            // waves per SIMD).  Rows that turn out to be irregular, empty or past the last read cost four wasted evaluations.
            double wq[NR][KP];
#pragma unroll
            for (int k = 0; k < KP; k++) {
                if (k * U.nt + wave * 64 < Wn) {                // (wave-uniform: a wave without placements in this pass skips it)
                    double px[NR], wx[NR];
#pragma unroll
                    for (int t = 0; t < NR; t++) px[t] = pc[t][k];
                    fig_weights_n<NR>(px, wx);
#pragma unroll
                    for (int t = 0; t < NR; t++) wq[t][k] = wx[t];
                } else {
#pragma unroll
                    for (int t = 0; t < NR; t++) wq[t][k] = 0.0;
                }
            }
#pragma unroll
            for (int t = 0; t < NR; t++) {
                const int s = q0 + t, r = c0 + s;
                if (r >= nU) continue;
                const int lo = __builtin_amdgcn_readlane(g_lo, t), hi = __builtin_amdgcn_readlane(g_hi, t);
                double *wrow = W + (long long)t * Wcap + (U.L - 1);
                const bool isirr = (irr >> t) & 1u;
                if (hi < lo) continue;
                if (!isirr) {
                    // weights of this read's placements: w = exp(0.5 log10 p) inside the read's window, 0 outside (:3591-3601);
                    // the wave's maximal product (a non-negative double orders like its bit pattern) and one offset that
                    // reaches it go to wv_v / wv_o
                    double pr = 0.0; int po = FIG_NOPOS;
#pragma unroll
                    for (int k = 0; k < KP; k++) {
                        const int i = tid + k * U.nt, o = i - (U.L - 1);
                        if (k * U.nt + wave * 64 < Wn) {        // (wave-uniform: a wave without placements in this pass skips it)
                            const bool in = i < Wn && o >= lo && o <= hi;
                            if (i < Wn) wrow[o] = in ? wq[t][k] : 0.0;
                            if (in) {
                                fl_acc += 4ULL * (unsigned long long)U.L + (unsigned long long)fig_ovl(o, U.L, G);
                                if (pc[t][k] > pr) { pr = pc[t][k]; po = o; }
                            }
                        }
                    }
                    FIG_TICK(E, 20);
                    long long bits; memcpy(&bits, &pr, 8);
                    const unsigned hi32 = (unsigned)((unsigned long long)bits >> 32), lo32 = (unsigned)((unsigned long long)bits & 0xffffffffULL);
                    const unsigned mh = fig_wave_max_u32(hi32);
                    const unsigned long long ah = fig_ballot(hi32 == mh);
                    // (nearly always one lane holds the maximal high word: its low word is the maximum's)
                    const unsigned ml = (ah & (ah - 1)) == 0ULL ? (unsigned)fig_lane_read_i32((int)lo32, fig_ctz64(ah)) : fig_wave_max_u32(hi32 == mh ? lo32 : 0u);
                    const unsigned long long am = fig_ballot(po != FIG_NOPOS && hi32 == mh && lo32 == ml);
                    const bool any = (mh | ml) != 0u && am != 0ULL;            // a product of 0 is no placement: log10 = -inf never beats -DBL_MAX
                    const int ao = any ? fig_lane_read_i32(po, fig_ctz64(am)) : FIG_NOPOS;
                    if (lane == 0) {
                        const unsigned long long mb = ((unsigned long long)mh << 32) | ml;
                        double mv; memcpy(&mv, &mb, 8);
                        SL->wv_v[t * 8 + wave] = any ? mv : 0.0; SL->wv_o[t * 8 + wave] = ao;
                    }
                    FIG_TICK(E, 21);
                } else {
                    // generic chain (N bases / short read): all lanes over the read's window, zero outside it
                    FigReadS rs; rs.len = fig_u(SL->sh_len[s]); rs.rev = fig_u(SL->sh_aux[s]) & 1; rs.hasN = 1; rs.pos = 0; rs.woff = wbase + fig_u((int)SL->sh_woff[s]);
                    fig_cu32p pk = (fig_cu32p)(U.packed + rs.woff);
                    const int nw2 = (rs.len + 15) >> 4;
                    fig_cdp kt = (fig_cdp)(rs.rev ? U.kt_rev + 2 * (U.L - rs.len) : U.kt_fwd);
                    const double *Q4 = fig_q4_ptr<LDS>(E);
                    const int tis0 = fig_u(SL->sh_tis0[s]), dir = (fig_u(SL->sh_aux[s]) & 4) ? 1 : -1;
                    for (int i = -(U.L - 1) + tid; i < G; i += U.nt) if (i < lo || i > hi) wrow[i] = 0.0;
                    FigBest best; best.v = -FIG_DBL_MAX; best.o = FIG_NOPOS;
                    unsigned long long nplace = 0, nadd = 0;
                    for (int o = lo + tid; o <= hi; o += U.nt) {
                        const double pr = fig_hot_chain_e<LDS>(PQ, Q4, U.ncolE, pk, nw2, kt, rs.len, 0, false, o + U.xoff, U.insd[tis0 + dir * o]);
                        const double tl = fig_log10(pr);
                        if (tl > best.v) { best.v = tl; best.o = o; }
                        wrow[o] = fig_exp(0.5 * tl);
                        nplace++; nadd += fig_ovl(o, rs.len, G);
                    }
                    fl_acc += 4ULL * nplace * (unsigned long long)rs.len + nadd;
                    const double bv = fig_wave_max_dpp(best.o == FIG_NOPOS ? -FIG_DBL_MAX : best.v);
                    const unsigned long long am = fig_ballot(best.o != FIG_NOPOS && best.v == bv);
                    const int ao = am ? fig_lane_read_i32(best.o, fig_ctz64(am)) : FIG_NOPOS;
                    if (lane == 0) { SL->wv_v[t * 8 + wave] = bv; SL->wv_o[t * 8 + wave] = bv > -FIG_DBL_MAX ? ao : FIG_NOPOS; }
                }
            }
            FIG_TICK(E, 36);
            FIG_SYNC();
            FIG_TICK(E, 37);
            // the next group's products (their latency hides behind this group's column pass)
            if (q0 + NR < nrd) {
#pragma unroll
                for (int t = 0; t < NR; t++)
#pragma unroll
                    for (int k = 0; k < KP; k++) { const int i = tid + k * U.nt; pn[t][k] = (i < Wn && c0 + q0 + NR + t < nU) ? prow[(long long)(q0 + NR + t) * pst + i] : 0.0; }
            }
            // ---- per-read bookkeeping (:3680-3688): regular reads hold products in wv_v (maxlv = log10 of the maximum: the
            // same value the placement's own log10 gave), reads of the generic chain hold log10 values
            if (tid >= 64 && tid < 64 + NR) {
                const int t = tid - 64, s = q0 + t;
                if (c0 + s < nU) {
                    FigBest b; b.v = -FIG_DBL_MAX; b.o = FIG_NOPOS;
                    if (SL->sh_hi[s] >= SL->sh_lo[s]) for (int k = 0; k < nw; k++) { FigBest y; y.v = SL->wv_v[t * 8 + k]; y.o = SL->wv_o[t * 8 + k]; b = fig_best_merge(b, y); }
                    if (b.o != FIG_NOPOS) E.scr.maxlv[c0 + s] = b.v;   // (regular reads: the product; its log10 is taken per super-chunk below)
                    else { E.scr.maxlv[c0 + s] = 0; fig_atomic_add_i32(&S.invalid_count, 1); }
                    E.scr.hint_e[c0 + s] = b.o;
                }
            }
            FIG_TICK(E, 27);
            // ---- column pass over the group's rows (as fig_hot_estep: base-owning waves, register accumulators)
            uint32_t plv_nx = plb[lane];                       // row 0's position lists; row t + 1's are fetched behind row t's additions
            for (int t = 0; t < NR && c0 + q0 + t < nU; t++) {
                if (U.nw == 8) { if ((t + (wave >> 2)) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
                const uint32_t plv = plv_nx;
                if (t + 1 < NR) plv_nx = plb[(t + 1) * 64 + lane];
                const int lo = __builtin_amdgcn_readlane(g_lo, t), hi = __builtin_amdgcn_readlane(g_hi, t);
                if (hi < lo) continue;
                const double *wrow = W + (long long)t * Wcap + (U.L - 1);
                const uint32_t cw0 = (uint32_t)__builtin_amdgcn_readlane((int)plv, 0), cw1 = (uint32_t)__builtin_amdgcn_readlane((int)plv, 1);
                const int nA = (int)(cw0 & 255), nC = (int)((cw0 >> 8) & 255), nG = (int)((cw0 >> 16) & 255), nT_ = (int)(cw0 >> 24);
                const int n4 = (int)(cw1 & 255);
                const int dC = 2 + ((nA + 3) >> 2), dG = dC + ((nC + 3) >> 2), dT = dG + ((nG + 3) >> 2), d4 = dT + ((nT_ + 3) >> 2);
                const int n = pb_c == 0 ? nA : pb_c == 1 ? nC : pb_c == 2 ? nG : nT_;
                const int d0 = pb_c == 0 ? 2 : pb_c == 1 ? dC : pb_c == 2 ? dG : dT;
                const double *wl = wrow + pb_x0;
                int k4 = 0;
                FIG_TICK(E, 28); FIG_COUNT(E, 31, 1); FIG_COUNT(E, 32, n * CPL);
                for (; k4 + 1 <= (n >> 2); k4++) {
                    const uint32_t s4 = (uint32_t)__builtin_amdgcn_readlane((int)plv, d0 + k4);
                    const double *p0 = wl - (int)(s4 & 255), *p1 = wl - (int)((s4 >> 8) & 255), *p2 = wl - (int)((s4 >> 16) & 255), *p3 = wl - (int)(s4 >> 24);
                    if (CPL <= 4) {
                        double w0[CPL], w1[CPL], w2[CPL], w3[CPL];
#pragma unroll
                        for (int i = 0; i < CPL; i++) { w0[i] = FIG_LDV(p0 + i * 64); w1[i] = FIG_LDV(p1 + i * 64); w2[i] = FIG_LDV(p2 + i * 64); w3[i] = FIG_LDV(p3 + i * 64); }
#pragma unroll
                        for (int i = 0; i < CPL; i++) acc[i] += w0[i];
#pragma unroll
                        for (int i = 0; i < CPL; i++) acc[i] += w1[i];
#pragma unroll
                        for (int i = 0; i < CPL; i++) acc[i] += w2[i];
#pragma unroll
                        for (int i = 0; i < CPL; i++) acc[i] += w3[i];
                    } else {
                        double w0[CPL], w1[CPL];
#pragma unroll
                        for (int i = 0; i < CPL; i++) { w0[i] = FIG_LDV(p0 + i * 64); w1[i] = FIG_LDV(p1 + i * 64); }
#pragma unroll
                        for (int i = 0; i < CPL; i++) acc[i] += w0[i];
#pragma unroll
                        for (int i = 0; i < CPL; i++) { w0[i] = FIG_LDV(p2 + i * 64); acc[i] += w1[i]; }
#pragma unroll
                        for (int i = 0; i < CPL; i++) { w1[i] = FIG_LDV(p3 + i * 64); acc[i] += w0[i]; }
#pragma unroll
                        for (int i = 0; i < CPL; i++) acc[i] += w1[i];
                    }
                }
                FIG_TICK(E, 29);
                if (n & 3) {
                    // the last one to three positions: their loads go out together (the list's padding bytes are 0, a valid offset)
                    const uint32_t s4 = (uint32_t)__builtin_amdgcn_readlane((int)plv, d0 + k4);
                    const int rem = n & 3;
                    const double *p0 = wl - (int)(s4 & 255), *p1 = wl - (int)((s4 >> 8) & 255), *p2 = wl - (int)((s4 >> 16) & 255);
                    double w0[CPL], w1[CPL], w2[CPL];
#pragma unroll
                    for (int i = 0; i < CPL; i++) { w0[i] = FIG_LDV(p0 + i * 64); w1[i] = FIG_LDV(p1 + i * 64); w2[i] = FIG_LDV(p2 + i * 64); }
#pragma unroll
                    for (int i = 0; i < CPL; i++) acc[i] += w0[i];
                    if (rem > 1) {
#pragma unroll
                        for (int i = 0; i < CPL; i++) acc[i] += w1[i];
                    }
                    if (rem > 2) {
#pragma unroll
                        for (int i = 0; i < CPL; i++) acc[i] += w2[i];
                    }
                }
                if (n4 > 0) {
                    for (int x = tid; x < G; x += U.nt) {
                        double a = E.scr.cnt[4 * cg + x];
                        for (int k = 0; k < n4; k++) {
                            const uint32_t s4 = (uint32_t)__builtin_amdgcn_readlane((int)plv, d4 + (k >> 2));
                            a += wrow[x - (int)((s4 >> ((k & 3) * 8)) & 255)];
                        }
                        E.scr.cnt[4 * cg + x] = a;
                    }
                }
                FIG_TICK(E, 30);
            }
            if (U.nw == 8) __builtin_amdgcn_s_setprio(0);
            FIG_TICK(E, 38);
            FIG_SYNC();
            FIG_TICK(E, 39);
        }
        // maxlv of the super-chunk's regular reads: log10 of the maximal product, one read per thread
        if (tid < nrd) {
            const bool isirr = (SL->sh_irr[tid >> 5] >> (tid & 31)) & 1u;
            if (!isirr && E.scr.hint_e[c0 + tid] != FIG_NOPOS) E.scr.maxlv[c0 + tid] = fig_log10(E.scr.maxlv[c0 + tid]);
        }
    }
#pragma unroll
    for (int i = 0; i < CPL; i++) {
        const int x = pb_x0 + i * U.wsz;
        if (x < G) E.scr.cnt[pb_c * cg + x] = acc[i];
    }
    E.flops += fl_acc;
    FIG_SYNC();
}

// Can this E-step take the shared-factor form?  (LDS table, not tiled, >= 4 weight rows, the gap not clipped by the contig
// start, at most two placements per thread and row in phase B, at most 8 column tiles per wave.)
FIG_D bool fig_sh_applies(const FigEng &E, int cpl) {
    if (!E.sh_on || E.tiles > 0 || !E.pq_lds || E.nteams < 4 || cpl > 8 || cpl < 1) return false;
    if (!E.B->ustream || E.M->L < 32) return false;
    const FigState &S = *E.S;
    if (S.left < E.xoff) return false;
    const int Wn = S.G + E.M->L - 1;
    return Wn <= 2 * E.nt && (E.nw == 4 || E.nw == 8) && (long long)FIG_SH_SC * FIG_SH_C * E.Wcap <= (long long)E.B->capW;
}

#endif  // !FIG_EMU
#endif
