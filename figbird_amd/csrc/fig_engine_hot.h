// fig_engine_hot.h -- the unmapped-mode hot loops of placeReads / finalize for gfx950.
//
// 99 % of the reference's time is the two triple loops of placeReads (Figbird.cpp:3530-3689 E-step,
// :3732-3846 MLE pass): for every read, every placement, every base one FP64 factor.  Mapping:
//
//   E-step (fig_hot_estep)
//   * reads go in chunks of `nteams` (= weight rows that fit in LDS); a TEAM of T = nw/nteams waves takes one read,
//     lanes = placements, two per lane (pair rounds; the tail of a read is split evenly over the team's waves);
//   * everything that is the same for all placements of a read at chain step j -- the read base b_j (2-bit packed
//     words), {1-e[k], e[k]} -- is wave-uniform and comes in through SCALAR loads (constant address space ->
//     s_load_dwordx16, s_bfe), so the vector pipe only sees one LDS read of {P,Q}[b_j][x] (16 B, conflict-free:
//     lane i reads column o_i + j; the pair's second placement at a constant offset of the same address) and four
//     FP64 ops per step: t1 = P*(1-e); t2 = e*Q; f = t1+t2; p *= f   (no FMA: -ffp-contract=off, as the x86
//     reference); 8-step blocks, the loads of block i+1 issued before the arithmetic of block i;
//   * the per-placement weights land in LDS (one row per read, zero outside the insert-size window); after a
//     workgroup barrier the lanes switch to COLUMNS: each wave owns one base and a run of 64-column tiles and adds
//     the weights that cover its columns in exactly the reference's (read, placement) order -- accumulators stay in
//     registers for the whole E-step (no atomics, so countsGap is bit-identical to the CPU);
//   MLE pass (fig_hot_mle)
//   * reuses the table area for C[to][x] = errorTypeProbs[from_x][to] (or -1 = match); one read per wave, taken
//     dynamically, no workgroup barrier; exact pruning (all factors <= 1): the previous pass's best placement is
//     evaluated first (fig_mle_serial), a whole-read mismatch filter on the 2-bit packed consensus (xor + popcount)
//     then discards every placement that cannot reach that maximum, the few others are evaluated one by one; rounds
//     the filter cannot judge run the pruned pair chain; arg-max is "first maximum wins".
#ifndef FIG_ENGINE_HOT_H
#define FIG_ENGINE_HOT_H

FIG_D FigBest fig_wave_best(const FigEng &E, FigBest x) {
#ifdef FIG_EMU
    (void)E;
    return x;
#else
    (void)E;
    // DPP row shifts / broadcasts (row_shr:1,2,4,8, row_bcast:15, row_bcast:31): lane 63 ends up with the merge of all
    // lanes; lanes that receive nothing merge with an empty entry.  The merge is order-independent (larger v, then
    // smaller o), so the scan order does not matter.
#define FIG_DPP_BEST(ctrl, rmask) do { \
        long long b_; memcpy(&b_, &x.v, 8); \
        const int lo_ = __builtin_amdgcn_update_dpp(0, (int)(b_ & 0xffffffffLL), ctrl, rmask, 0xf, false); \
        const int hi_ = __builtin_amdgcn_update_dpp(0, (int)(b_ >> 32), ctrl, rmask, 0xf, false); \
        const int oo_ = __builtin_amdgcn_update_dpp(FIG_NOPOS, x.o, ctrl, rmask, 0xf, false); \
        FigBest y_; long long o2_ = ((long long)hi_ << 32) | (unsigned int)lo_; memcpy(&y_.v, &o2_, 8); y_.o = oo_; \
        x = fig_best_merge(x, y_); } while (0)
    FIG_DPP_BEST(0x111, 0xf); FIG_DPP_BEST(0x112, 0xf); FIG_DPP_BEST(0x114, 0xf); FIG_DPP_BEST(0x118, 0xf);
    FIG_DPP_BEST(0x142, 0xa); FIG_DPP_BEST(0x143, 0xc);
#undef FIG_DPP_BEST
    // callers read the result in lane 0
    long long b2; memcpy(&b2, &x.v, 8);
    const unsigned int l = (unsigned int)__builtin_amdgcn_readlane((int)(b2 & 0xffffffffLL), 63), h = (unsigned int)__builtin_amdgcn_readlane((int)(b2 >> 32), 63);
    b2 = (long long)(((unsigned long long)h << 32) | l);
    memcpy(&x.v, &b2, 8);
    x.o = __builtin_amdgcn_readlane(x.o, 63);
    return x;                         // valid in every lane
#endif
}

// FigEng lives in private memory once the big control functions are not inlined, so the compiler no longer
// knows that its fields are wave-uniform.  The hot loops re-assert uniformity with readfirstlane, which
// turns the dependent address arithmetic back into SALU and the table loads into s_load.
FIG_D int fig_u(int v) { return FIG_RFL(v); }
template <class T> FIG_D T *fig_uptr(T *p) {
#ifdef FIG_EMU
    return p;
#else
    unsigned long long v = (unsigned long long)p;
    unsigned lo = FIG_RFL((unsigned)(v & 0xffffffffu)), hi = FIG_RFL((unsigned)(v >> 32));
    return (T *)(((unsigned long long)hi << 32) | lo);
#endif
}
FIG_D long long fig_u64(long long w) {
#ifdef FIG_EMU
    return w;
#else
    unsigned lo = FIG_RFL((unsigned)((unsigned long long)w & 0xffffffffu)), hi = FIG_RFL((unsigned)((unsigned long long)w >> 32));
    return (long long)(((unsigned long long)hi << 32) | lo);
#endif
}

// Wave maximum through DPP row shifts / broadcasts (gfx9 row_shr:1,2,4,8 then row_bcast:15 and :31): three vector
// instructions per step and no LDS traffic, against three ds_bpermute per step of the shuffle form.  Result in every lane.
FIG_D double fig_wave_max_dpp(double v) {
#ifdef FIG_EMU
    return v;
#else
    const double ident = -FIG_DBL_MAX * 2.0;          // -inf: lanes that receive nothing keep their own value through max
#define FIG_DPP_STEP(ctrl, rmask) do { \
        long long b_; memcpy(&b_, &v, 8); long long i_; memcpy(&i_, &ident, 8); \
        int lo_ = __builtin_amdgcn_update_dpp((int)(i_ & 0xffffffffLL), (int)(b_ & 0xffffffffLL), ctrl, rmask, 0xf, false); \
        int hi_ = __builtin_amdgcn_update_dpp((int)(i_ >> 32), (int)(b_ >> 32), ctrl, rmask, 0xf, false); \
        long long o_ = ((long long)hi_ << 32) | (unsigned int)lo_; double y_; memcpy(&y_, &o_, 8); \
        v = y_ > v ? y_ : v; } while (0)
    FIG_DPP_STEP(0x111, 0xf); FIG_DPP_STEP(0x112, 0xf); FIG_DPP_STEP(0x114, 0xf); FIG_DPP_STEP(0x118, 0xf);
    FIG_DPP_STEP(0x142, 0xa); FIG_DPP_STEP(0x143, 0xc);
#undef FIG_DPP_STEP
    long long b2; memcpy(&b2, &v, 8);
    const unsigned int l = (unsigned int)__builtin_amdgcn_readlane((int)(b2 & 0xffffffffLL), 63), h = (unsigned int)__builtin_amdgcn_readlane((int)(b2 >> 32), 63);
    b2 = (long long)(((unsigned long long)h << 32) | l);
    memcpy(&v, &b2, 8);
    return v;
#endif
}

FIG_D int fig_ctz64(unsigned long long m) { return __builtin_ctzll(m); }   // m != 0
FIG_D int fig_lane_read_i32(int v, int lane) {      // value of `v` in lane `lane` (wave-uniform index)
#ifdef FIG_EMU
    (void)lane;
    return v;
#else
    return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(lane));
#endif
}
FIG_D unsigned long long fig_ballot(bool p) {
#ifdef FIG_EMU
    return p ? 1ULL : 0ULL;
#else
    return __ballot(p);
#endif
}
FIG_D double fig_bcast_d(double v) {       // lane 0's value in every lane
    long long bits; memcpy(&bits, &v, 8); bits = fig_u64(bits); memcpy(&v, &bits, 8);
    return v;
}

FIG_D bool fig_wave_any(bool p) {
#ifdef FIG_EMU
    return p;
#else
    return __any(p) != 0;
#endif
}

template <bool LDS> FIG_D const FigPQ *fig_pq_ptr(const FigEng &E) { return LDS ? (const FigPQ *)(fig_lds + fig_u(E.off_pq)) : fig_uptr(E.pq); }
template <bool LDS> FIG_D const double *fig_q4_ptr(const FigEng &E) { return LDS ? (const double *)(fig_lds + fig_u(E.off_q4)) : fig_uptr(E.q4); }
template <bool LDS> FIG_D double *fig_w_ptr(const FigEng &E) { return LDS ? (double *)(fig_lds + fig_u(E.off_w)) : fig_uptr(E.wbuf); }
template <bool LDS> FIG_D double *fig_c_ptr(const FigEng &E) { return LDS ? (double *)(fig_lds + fig_u(E.off_pq)) : (double *)fig_uptr(E.pq); }

// volatile keeps the column-pass loads as single ds_read_b64 (256 B/clk); merged ds_read2st64_b64 runs at half that rate
#ifdef FIG_EMU
#define FIG_LDV(p) (*(p))
#else
template <bool LDS> FIG_D double fig_ldv(const double *p) {
    if (LDS) return *(const volatile __attribute__((address_space(3))) double *)p;
    return *p;
}
#define FIG_LDV(p) fig_ldv<LDS>(p)
#endif

// base j of a packed read through scalar loads
FIG_D int fig_sbase(fig_cu32p pk, int nw2, int j) {
    uint32_t w = pk[j >> 4], m = pk[nw2 + (j >> 5)];
    return ((m >> (j & 31)) & 1) ? 4 : (int)((w >> ((j & 15) * 2)) & 3);
}

// E-step product chain of one placement (one lane).  xe0 = o + xoff.
template <bool LDS>
FIG_D double fig_hot_chain_e(const FigPQ *PQ, const double *Q4, int ncolE, fig_cu32p pk, int nw2, fig_cdp kt, int len,
                             int jstart, bool clipped, int xe0, double p) {
    for (int wi = 0; wi < nw2; wi++) {
        uint32_t w = pk[wi];
        uint32_t m = (pk[nw2 + (wi >> 1)] >> ((wi & 1) * 16)) & 0xffffu;
        int j0 = wi * 16;
        int nb = len - j0; if (nb > 16) nb = 16;
        fig_cdp k2 = kt + 2 * j0;
        const FigPQ *col = PQ + xe0 + j0;
        if (m == 0 && nb == 16 && !clipped) {
            // issue the 16 LDS reads of the block first, then run the dependent multiply chain
            FigPQ v[16];
#pragma unroll
            for (int jj = 0; jj < 16; jj++) v[jj] = col[(int)((w >> (2 * jj)) & 3) * ncolE + jj];
#pragma unroll
            for (int jj = 0; jj < 16; jj++) {
                double ome = k2[2 * jj], e = k2[2 * jj + 1];
                p *= (v[jj].p * ome + e * v[jj].q);
            }
        } else {
            for (int jj = 0; jj < nb; jj++) {
                if (j0 + jj < jstart) continue;
                double ome = k2[2 * jj], e = k2[2 * jj + 1];
                if ((m >> jj) & 1) p *= (e * Q4[xe0 + j0 + jj]);
                else {
                    int b = (int)((w >> (2 * jj)) & 3);
                    FigPQ v = col[b * ncolE + jj];
                    p *= (v.p * ome + e * v.q);
                }
            }
        }
    }
    return p;
}

// Two placements per lane at once (same read => same scalar data): twice the independent work per
// scalar-table fetch and two independent multiply chains to hide FP64 latency.  The chain is cut into 8-step
// blocks; each block first issues its scalar table loads (32 SGPRs) and its 16 LDS reads, then runs the 64 FP64
// ops, so the memory latencies overlap each other and (with a second wave on the SIMD) the arithmetic.
struct FigEBlk { double kk[16]; FigPQ va[8], vb[8]; };

// DX > 0: the second placement sits DX columns right of the first (xb == xa + DX), so both reads of a step share one
// address register and the second uses the instruction's constant offset.  DX == 0: independent xb.
// The last block of a chain may be partial: its loads run past the read's end (2-bit codes there are 0 and the
// table / weight-pair arrays continue, so the addresses are valid) and only the first `n` steps are multiplied in.
template <bool LDS, int DX>
FIG_D void fig_eblk_load(FigEBlk &B, const FigPQ *PQ, int ncolE, fig_cu32p pk, fig_cdp kt, int bi, int xa, int xb) {
    const int j0 = bi * 8;
    const uint32_t w = pk[bi >> 1] >> ((bi & 1) * 16);
    fig_cdp k2 = kt + 2 * j0;
#pragma unroll
    for (int q = 0; q < 16; q++) B.kk[q] = k2[q];
    const FigPQ *ca = PQ + xa + j0, *cb = DX > 0 ? ca + DX : PQ + xb + j0;
#pragma unroll
    for (int jj = 0; jj < 8; jj++) { int r = (int)((w >> (2 * jj)) & 3) * ncolE + jj; B.va[jj] = ca[r]; B.vb[jj] = cb[r]; }
}
FIG_D void fig_eblk_compute(const FigEBlk &B, double &pa, double &pb) {
#pragma unroll
    for (int jj = 0; jj < 8; jj++) {
        pa *= (B.va[jj].p * B.kk[2 * jj] + B.kk[2 * jj + 1] * B.va[jj].q);
        pb *= (B.vb[jj].p * B.kk[2 * jj] + B.kk[2 * jj + 1] * B.vb[jj].q);
    }
}
FIG_D void fig_eblk_compute_n(const FigEBlk &B, int n, double &pa, double &pb) {      // n in 1..8, wave-uniform
#pragma unroll
    for (int jj = 0; jj < 8; jj++) {
        if (jj < n) {
            pa *= (B.va[jj].p * B.kk[2 * jj] + B.kk[2 * jj + 1] * B.va[jj].q);
            pb *= (B.vb[jj].p * B.kk[2 * jj] + B.kk[2 * jj + 1] * B.vb[jj].q);
        }
    }
}

// Software-pipelined: the loads of block i+1 (scalar table + 16 LDS reads) are issued before the 64 FP64 ops
// of block i, so their latency is covered by arithmetic of the same wave.
template <bool LDS, int DX>
FIG_D void fig_hot_chain_e2(const FigPQ *PQ, int ncolE, fig_cu32p pk, int nw2, fig_cdp kt, int len, int xa, int xb, double &pa, double &pb) {
    (void)nw2;
    const int nblk = (len + 7) >> 3;                 // the last one holds len - 8*(nblk-1) steps
    const int nlast = len - 8 * (nblk - 1);
    if (nblk <= 0) return;
    FigEBlk A, Bk;
    fig_eblk_load<LDS, DX>(A, PQ, ncolE, pk, kt, 0, xa, xb);
    int bi = 0;
    for (; bi + 2 <= nblk - 1; bi += 2) {
        fig_eblk_load<LDS, DX>(Bk, PQ, ncolE, pk, kt, bi + 1, xa, xb);
        fig_eblk_compute(A, pa, pb);
        fig_eblk_load<LDS, DX>(A, PQ, ncolE, pk, kt, bi + 2, xa, xb);
        fig_eblk_compute(Bk, pa, pb);
    }
    // here block `bi` is loaded in A; blocks bi+1 .. nblk-1 remain (0 or 1 of them)
    if (bi + 1 <= nblk - 1) {
        fig_eblk_load<LDS, DX>(Bk, PQ, ncolE, pk, kt, bi + 1, xa, xb);
        fig_eblk_compute(A, pa, pb);
        fig_eblk_compute_n(Bk, nlast, pa, pb);
    } else fig_eblk_compute_n(A, nlast, pa, pb);
}

// One placement per lane, same pipeline (the short last round of a read).
struct FigEBlk1 { double kk[16]; FigPQ va[8]; };
template <bool LDS>
FIG_D void fig_eblk1_load(FigEBlk1 &B, const FigPQ *PQ, int ncolE, fig_cu32p pk, fig_cdp kt, int bi, int xa) {
    const int j0 = bi * 8;
    const uint32_t w = pk[bi >> 1] >> ((bi & 1) * 16);
    fig_cdp k2 = kt + 2 * j0;
#pragma unroll
    for (int q = 0; q < 16; q++) B.kk[q] = k2[q];
    const FigPQ *ca = PQ + xa + j0;
#pragma unroll
    for (int jj = 0; jj < 8; jj++) B.va[jj] = ca[(int)((w >> (2 * jj)) & 3) * ncolE + jj];
}
FIG_D void fig_eblk1_compute_n(const FigEBlk1 &B, int n, double &pa) {
#pragma unroll
    for (int jj = 0; jj < 8; jj++) if (jj < n) pa *= (B.va[jj].p * B.kk[2 * jj] + B.kk[2 * jj + 1] * B.va[jj].q);
}
template <bool LDS>
FIG_D void fig_hot_chain_e1(const FigPQ *PQ, int ncolE, fig_cu32p pk, fig_cdp kt, int len, int xa, double &pa) {
    const int nblk = (len + 7) >> 3;
    const int nlast = len - 8 * (nblk - 1);
    if (nblk <= 0) return;
    FigEBlk1 A, Bk;
    fig_eblk1_load<LDS>(A, PQ, ncolE, pk, kt, 0, xa);
    int bi = 0;
    for (; bi + 2 <= nblk - 1; bi += 2) {
        fig_eblk1_load<LDS>(Bk, PQ, ncolE, pk, kt, bi + 1, xa);
        fig_eblk1_compute_n(A, 8, pa);
        fig_eblk1_load<LDS>(A, PQ, ncolE, pk, kt, bi + 2, xa);
        fig_eblk1_compute_n(Bk, 8, pa);
    }
    if (bi + 1 <= nblk - 1) {
        fig_eblk1_load<LDS>(Bk, PQ, ncolE, pk, kt, bi + 1, xa);
        fig_eblk1_compute_n(A, 8, pa);
        fig_eblk1_compute_n(Bk, nlast, pa);
    } else fig_eblk1_compute_n(A, nlast, pa);
}

struct FigMBlk { double kk[16]; double va[8], vb[8]; };

template <bool LDS, int DX>
FIG_D void fig_mblk_load(FigMBlk &B, const double *C, int ncolE, fig_cu32p pk, fig_cdp mt, int bi, int xa, int xb) {
    const int j0 = bi * 8;
    const uint32_t w = pk[bi >> 1] >> ((bi & 1) * 16);
    fig_cdp k2 = mt + 2 * j0;
#pragma unroll
    for (int q = 0; q < 16; q++) B.kk[q] = k2[q];
    const double *ca = C + xa + j0, *cb = DX > 0 ? ca + DX : C + xb + j0;
#pragma unroll
    for (int jj = 0; jj < 8; jj++) { int r = (int)((w >> (2 * jj)) & 3) * ncolE + jj; B.va[jj] = ca[r]; B.vb[jj] = cb[r]; }
}
FIG_D void fig_mblk_compute(const FigMBlk &B, double &qa, double &qb) {
#pragma unroll
    for (int jj = 0; jj < 8; jj++) {
        const double fa = B.kk[2 * jj + 1] * B.va[jj], fb = B.kk[2 * jj + 1] * B.vb[jj];
        qa *= (B.va[jj] < 0 ? B.kk[2 * jj] : fa);
        qb *= (B.vb[jj] < 0 ? B.kk[2 * jj] : fb);
    }
}
FIG_D void fig_mblk_compute_n(const FigMBlk &B, int n, double &qa, double &qb) {     // n in 1..8, wave-uniform
#pragma unroll
    for (int jj = 0; jj < 8; jj++) {
        if (jj < n) {
            const double fa = B.kk[2 * jj + 1] * B.va[jj], fb = B.kk[2 * jj + 1] * B.vb[jj];
            qa *= (B.va[jj] < 0 ? B.kk[2 * jj] : fa);
            qb *= (B.vb[jj] < 0 ? B.kk[2 * jj] : fb);
        }
    }
}

// Pruned MLE pair chain, software-pipelined like fig_hot_chain_e2 (incl. the partial last block and DX).  All factors
// are <= 1, so partial products only shrink: a lane is dead once its product is below `bound`.  After every block:
//   no live lane            -> return 0 (nothing in this round can reach the bound);
//   `few` live lanes, and finishing them one by one (fig_mle_serial, ~4 blocks' time each) is cheaper than running
//   the remaining blocks    -> return 1 with their lane masks in ma / mb;
//   otherwise continue; return 2 with the finished products in qa / qb.
// The prefetched block of an early return is simply dropped.  With DX > 0 a lane whose second placement is outside
// the window (vb_ok false) still reads xa + DX: inside the table area, result ignored.
FIG_D int fig_popc64(unsigned long long m) { int c = 0; for (; m; m &= m - 1) c++; return c; }
#define FIG_M2P_CHECK(done) do { \
        ma = fig_ballot(va_ok && qa >= bound); mb = fig_ballot(vb_ok && qb >= bound); \
        nb_done = (done); \
        if (!(ma | mb)) return 0; \
        if (allow_few) { const int alive_ = fig_popc64(ma) + fig_popc64(mb); if (alive_ <= 8 && (nblk - (done)) > 4 * alive_) return 1; } \
    } while (0)
template <bool LDS, int DX>
FIG_D int fig_hot_chain_m2p(const double *C, int ncolE, fig_cu32p pk, fig_cdp mt, int len, int xa, int xb, bool va_ok, bool vb_ok,
                            double bound, bool allow_few, double &qa, double &qb, unsigned long long &ma, unsigned long long &mb, int &nb_done) {
    const int nblk = (len + 7) >> 3;
    const int nlast = len - 8 * (nblk - 1);
    ma = mb = 0; nb_done = nblk > 0 ? nblk : 0;
    if (nblk <= 0) return 2;
    FigMBlk A, Bk;
    fig_mblk_load<LDS, DX>(A, C, ncolE, pk, mt, 0, xa, xb);
    int bi = 0;
    for (; bi + 2 <= nblk - 1; bi += 2) {
        fig_mblk_load<LDS, DX>(Bk, C, ncolE, pk, mt, bi + 1, xa, xb);
        fig_mblk_compute(A, qa, qb);
        FIG_M2P_CHECK(bi + 1);
        fig_mblk_load<LDS, DX>(A, C, ncolE, pk, mt, bi + 2, xa, xb);
        fig_mblk_compute(Bk, qa, qb);
        FIG_M2P_CHECK(bi + 2);
    }
    if (bi + 1 <= nblk - 1) {
        fig_mblk_load<LDS, DX>(Bk, C, ncolE, pk, mt, bi + 1, xa, xb);
        fig_mblk_compute(A, qa, qb);
        FIG_M2P_CHECK(bi + 1);
        fig_mblk_compute_n(Bk, nlast, qa, qb);
    } else fig_mblk_compute_n(A, nlast, qa, qb);
    nb_done = nblk;
    return 2;
}

// MLE product chain of one placement.  C[to*ncolE + xe] = -1 when the consensus base equals `to`,
// else errorTypeProbs[from][to].  mt = {1-e-ins-del, e} pairs.
template <bool LDS>
FIG_D double fig_hot_chain_m(const double *C, int ncolE, fig_cu32p pk, int nw2, fig_cdp mt, int len, int xe0) {
    double q = 1;
    for (int wi = 0; wi < nw2; wi++) {
        uint32_t w = pk[wi];
        uint32_t m = (pk[nw2 + (wi >> 1)] >> ((wi & 1) * 16)) & 0xffffu;
        int j0 = wi * 16;
        int nb = len - j0; if (nb > 16) nb = 16;
        fig_cdp k2 = mt + 2 * j0;
        const double *col = C + xe0 + j0;
        if (m == 0 && nb == 16) {
            double cv[16];
#pragma unroll
            for (int jj = 0; jj < 16; jj++) cv[jj] = col[(int)((w >> (2 * jj)) & 3) * ncolE + jj];
#pragma unroll
            for (int jj = 0; jj < 16; jj++) {
                double m3 = k2[2 * jj], e = k2[2 * jj + 1];
                double f = e * cv[jj];
                q *= (cv[jj] < 0 ? m3 : f);
            }
        } else {
            for (int jj = 0; jj < nb; jj++) {
                int b = ((m >> jj) & 1) ? 4 : (int)((w >> (2 * jj)) & 3);
                double m3 = k2[2 * jj], e = k2[2 * jj + 1];
                double c = col[b * ncolE + jj];
                double f = e * c;
                q *= (c < 0 ? m3 : f);
            }
        }
    }
    return q;
}

// Table of the MLE pass over the extended columns, from the consensus codes E.gs[0..G) and the
// flanks (gapString of Figbird.cpp:3304-3323 / :4978-4996).  Overlays the {P,Q} table, which every
// caller rebuilds (computeProbsGap) before the next E-step.  All lanes; ends with a barrier.
FIG_D void fig_build_mle_table(FigEng &E, int G, int left, int right) {
    const double *T = E.M->T;
    double *C = (double *)E.pq;
    int n = G + 2 * E.xoff;
    if (n > E.ncolE) n = E.ncolE;
    const fig_gcu8p fl = (fig_gcu8p)fig_uptr(E.B)->flank + fig_u64(fig_uptr(E.g)->flankOff);
    const fig_lcu8p gsl = (fig_lcu8p)E.gs;
    for (int i = E.tid; i < n; i += E.nt) {
        int x = i - E.xoff;
        int from = fig_from_code_u(fl, gsl, x, G, left, right);
        for (int to = 0; to < 5; to++) C[(long long)to * E.ncolE + i] = (from == to) ? -1.0 : T[from * 5 + to];
    }
    FIG_SYNC();
}

struct FigReadS { int len, rev, hasN, pos; long long woff; };
struct alignas(16) FigU4 { uint32_t x, y, z, w; };      // one ds_read_b128

struct FigHotU {                     // wave-uniform copies of what the hot loops need
    const int32_t *u_len, *u_aux, *u_pos; const int64_t *u_woff; const uint32_t *packed; const double *insd;
    const double *kt_fwd, *kt_rev, *mt_fwd, *mt_rev;
    long long ub, gapStart;
    int G, left, nU, cg, ncolE, xoff, Wcap, nteams, nw, wsz, L, Tmin, Tmax, cutoff, G0, nt;
};
FIG_D FigHotU fig_hot_uniforms(const FigEng &E) {
    FigHotU U;
    const FigDevBatch *B = fig_uptr(E.B);
    const FigDevModel *M = fig_uptr(E.M);
    const FigDevGap *g = fig_uptr(E.g);
    const FigState *S = E.S;
    U.u_len = fig_uptr(B->u.len); U.u_aux = fig_uptr(B->u.aux); U.u_pos = fig_uptr(B->u.pos); U.u_woff = fig_uptr(B->u.woff);
    U.packed = fig_uptr(B->packed); U.insd = fig_uptr(M->insd);
    U.kt_fwd = fig_uptr(E.kt_fwd); U.kt_rev = fig_uptr(E.kt_rev); U.mt_fwd = fig_uptr(E.mt_fwd); U.mt_rev = fig_uptr(E.mt_rev);
    U.ub = fig_u64(g->uBase); U.gapStart = fig_u64(g->gapStart);
    U.G = fig_u(S->G); U.left = fig_u(S->left); U.nU = fig_u(g->nU); U.cg = fig_u(E.capG); U.ncolE = fig_u(E.ncolE); U.xoff = fig_u(E.xoff);
    U.Wcap = fig_u(E.Wcap); U.nteams = fig_u(E.nteams); U.nw = fig_u(E.nw); U.wsz = fig_u(E.wsz); U.L = fig_u(M->L);
    U.Tmin = fig_u(M->Tmin); U.Tmax = fig_u(M->Tmax); U.cutoff = fig_u(M->cutoff); U.G0 = fig_u(g->G0); U.nt = fig_u(E.nt);
    return U;
}
FIG_D FigReadS fig_read_scalars(const FigHotU &U, long long idx) {
    FigReadS r;
    r.len = fig_u(U.u_len[idx]);
    { const int a = fig_u(U.u_aux[idx]); r.rev = a & 1; r.hasN = (a >> 1) & 1; }
    r.pos = fig_u(U.u_pos[idx]);
    r.woff = fig_u64(U.u_woff[idx]);
    return r;
}
FIG_D FigWin fig_window_u(const FigHotU &U, int pos1, int len, int gapoffset) {     // fig_window_unmapped on uniforms
    FigWin w;
    int lo = -(len - 1), hi = U.G - 1;
    if ((long long)pos1 < U.gapStart) {
        w.tis0 = (int)(U.gapStart - pos1 + len); w.dir = 1;
        int a = U.Tmin - w.tis0, b = U.Tmax - w.tis0;
        if (a > lo) lo = a;
        if (b < hi) hi = b;
    } else {
        int p = pos1 + gapoffset;
        w.tis0 = (int)(p - U.gapStart + len); w.dir = -1;
        int a = w.tis0 - U.Tmax, b = w.tis0 - U.Tmin;
        if (a > lo) lo = a;
        if (b < hi) hi = b;
    }
    w.lo = lo; w.hi = hi;
    return w;
}

// LDS-tiled class: copy columns [x0, x0 + tcols) of the L2-resident table {P,Q}[4][ncolE] + Q4[ncolE] (E.pq, E.q4) into
// the LDS image {P,Q}[4][tcols] + Q4[tcols] at off_pq / off_q4.  All lanes; the caller brackets it with barriers.
// Columns past the table's end are left as they are: no placement of a valid lane reaches them.
FIG_D void fig_tile_copy(FigEng &E, int x0, int tcols) {
    const int ncolE = fig_u(E.ncolE);
    const FigPQ *src = fig_uptr(E.pq) + x0; const double *sq = fig_uptr(E.q4) + x0;
    FigPQ *dst = (FigPQ *)(fig_lds + fig_u(E.off_pq)); double *dq = fig_lds + fig_u(E.off_q4);
    int n = ncolE - x0; if (n > tcols) n = tcols;
    // the five loads of a column are issued together (the copy is latency-bound: one L2 round trip per loop trip)
    for (int x = E.tid; x < n; x += E.nt) {
        const FigPQ v0 = src[x], v1 = src[ncolE + x], v2 = src[2 * ncolE + x], v3 = src[3 * ncolE + x];
        const double q = sq[x];
        dst[x] = v0; dst[tcols + x] = v1; dst[2 * tcols + x] = v2; dst[3 * tcols + x] = v3; dq[x] = q;
    }
}

// ---------------------------------------------------------------------------------------
// E-step over all unmapped reads of the gap (Figbird.cpp:3530-3689).  CPL = columns per lane.
template <bool LDS, int CPL, bool TILED>
FIG_NOINLINE FIG_D void fig_hot_estep(FigEng &E, int gapoffset) {
    FigState &S = *E.S;
    const FigHotU U = fig_hot_uniforms(E);
    const int G = U.G, left = U.left, nU = U.nU, cg = U.cg;
    const long long ub = U.ub;
    const FigPQ *PQ = fig_pq_ptr<LDS>(E);
    const double *Q4 = fig_q4_ptr<LDS>(E);
    double *W = fig_w_ptr<LDS>(E);
    const int ncolE = U.ncolE, xoff = U.xoff, Wcap = U.Wcap;
    int nteams = U.nteams; if (nteams > U.nw) nteams = U.nw;
    const int T = U.nw / nteams;
    const int wave = fig_u(E.wave), lane = E.lane, tid = E.tid;
    const int team = wave / T, wit = wave - team * T;
    const bool clipped = left < xoff;              // some placements start left of the window (gap near the contig start)
    // LDS-tiled class (table too wide for LDS): PQ/Q4 above point at an image of tile_cols columns, refilled per tile
    const int ntl = TILED ? fig_u(E.tiles) : 1, tstep = TILED ? fig_u(E.tile_step) : 0;
    const int pstride = TILED ? fig_u(E.tile_cols) : ncolE;      // column stride of the image the chains read
    const int Tm = TILED ? U.nw : T;                             // arg-max slots per read (S.wv_*)
    // Column pass: wave w owns base c = w & 3 and CPL consecutive 64-column tiles starting at tile (w >> 2) * CPL,
    // one register accumulator per tile for the whole E-step.  All tiles of a wave walk the same position list
    // (that base's positions in the read, four per staged dword), so the list decode is shared and the CPL add
    // chains are independent.  Reads with N bases add their N positions straight into countsGap[4].
    // CPL == 0: generic fallback (very long gaps, and the one-lane CPU emulation): accumulate in E.scr.cnt.
    double acc[CPL > 0 ? CPL : 1];
#pragma unroll
    for (int m = 0; m < (CPL > 0 ? CPL : 1); m++) acc[m] = 0;
    const int pb_c = wave & 3, pb_x0 = (wave >> 2) * CPL * U.wsz + lane;       // this wave's base and first column
    unsigned long long fl_acc = 0;                   // this lane's flops of the call: FigEng sits in private memory, one update per call
#ifndef FIG_EMU
    uint32_t *plb = (uint32_t *)(fig_lds + fig_u(E.off_plb));
#else
    uint32_t *plb = E.plb;
#endif
    FIG_SYNC();                                       // placeReads zeroed countsGap already (:3050-3056)

    FIG_T0(E);
    for (int c0 = 0; c0 < nU; c0 += nteams) {
        if (!TILED) {
            // ---- phase A: lanes = placements
            int r = c0 + team;
            FigBest best; best.v = -FIG_DBL_MAX; best.o = FIG_NOPOS;
            if (team < nteams && r < nU) {
                FIG_T0(E);
                FigReadS rs = fig_read_scalars(U, ub + r);
                FigWin w = fig_window_u(U, rs.pos, rs.len, gapoffset);
                if (wit == 0 && lane == 0) { S.tm_lo[team] = w.lo; S.tm_hi[team] = w.hi; S.tm_len[team] = rs.len; }
                fig_cu32p pk = (fig_cu32p)(U.packed + rs.woff);
                int nw2 = (rs.len + 15) >> 4;
                // stage {counts, position lists} of this read for the column pass: one dword per lane, stored below
                uint32_t plv = 0;
                if (CPL > 0 && wit == 0) { const int ndw = 2 + ((rs.len + 3) >> 2) + 4; if (lane < ndw) plv = pk[nw2 + ((rs.len + 31) >> 5) + lane]; }
                fig_cdp kt = (fig_cdp)(rs.rev ? U.kt_rev + 2 * (U.L - rs.len) : U.kt_fwd);
                // weight row covers every placement offset o in [-(L-1), G-1] at index o+(L-1); offsets outside the
                // insert-size window hold 0.0 so that the column pass needs no window test (x + 0.0 == x exactly)
                double *wrow = W + (long long)team * Wcap + (U.L - 1);
                for (int i = -(U.L - 1) + wit * U.wsz + lane; i < G; i += T * U.wsz) if (i < w.lo || i > w.hi) wrow[i] = 0.0;
                unsigned long long nplace = 0, nadd = 0;
                // does the read contain an N?  (then the generic chain handles it)
                const bool hasN = rs.hasN != 0;
                const int stride = T * U.wsz;
                int o = w.lo + wit * U.wsz + lane;
                if (!hasN && !clipped) {
                    // Wave-uniform rounds.  Full team rounds (2*stride placements: every wave of the team has both halves of
                    // its pair), then the tail of < 2*stride placements is cut into T equal stretches, one per wave, so the
                    // waves of a team finish together: a stretch of more than one wave width runs as a pair round whose second
                    // half is masked (those lanes read inside the LDS image and are ignored), a shorter one as a single round.
                    const int Wn = w.hi - w.lo + 1;
                    const int nfull = Wn > 0 ? Wn / (2 * stride) : 0;
                    // insert-size terms of the next round are fetched before the current round's chains (their latency would
                    // otherwise sit in front of the first multiply)
                    double pa_n = 0, pb_n = 0;
                    if (nfull > 0) { const int oa0 = w.lo + wit * U.wsz + lane; pa_n = U.insd[w.tis0 + w.dir * oa0]; pb_n = U.insd[w.tis0 + w.dir * (oa0 + stride)]; }
                    FIG_TICK(E, 22);
                    for (int k = 0; k < nfull; k++) {
#ifndef FIG_EMU
                        // 8-wave workgroups put waves w and w+4 on one SIMD and the arbiter serves the older one first; taking
                        // turns at the higher priority, round by round, lets the two finish the phase together
                        if (U.nw == 8) { if ((k + (wave >> 2)) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
#endif
                        const int oa = w.lo + k * 2 * stride + wit * U.wsz + lane, ob = oa + stride;
                        double pa = pa_n, pb = pb_n;
                        if (k + 1 < nfull) { pa_n = U.insd[w.tis0 + w.dir * (oa + 2 * stride)]; pb_n = U.insd[w.tis0 + w.dir * (ob + 2 * stride)]; }
                        FIG_T0(E);
#ifdef FIG_EMU
                        fig_hot_chain_e2<LDS, 0>(PQ, ncolE, pk, nw2, kt, rs.len, oa + xoff, ob + xoff, pa, pb);
#else
                        if (T == 1) fig_hot_chain_e2<LDS, 64>(PQ, ncolE, pk, nw2, kt, rs.len, oa + xoff, 0, pa, pb);
                        else if (T == 2) fig_hot_chain_e2<LDS, 128>(PQ, ncolE, pk, nw2, kt, rs.len, oa + xoff, 0, pa, pb);
                        else fig_hot_chain_e2<LDS, 0>(PQ, ncolE, pk, nw2, kt, rs.len, oa + xoff, ob + xoff, pa, pb);
#endif
                        FIG_TICK(E, 9);
                        const double ta = fig_log10(pa), tb = fig_log10(pb);
                        if (ta > best.v) { best.v = ta; best.o = oa; }
                        if (tb > best.v) { best.v = tb; best.o = ob; }
                        wrow[oa] = fig_exp(0.5 * ta);
                        wrow[ob] = fig_exp(0.5 * tb);
                        FIG_TICK(E, 10);
                        nplace += 2; nadd += fig_ovl(oa, rs.len, G) + fig_ovl(ob, rs.len, G);
                    }
#ifndef FIG_EMU
                    if (U.nw == 8) __builtin_amdgcn_s_setprio(0);
#endif
                    {
                        const int tbase = w.lo + nfull * 2 * stride;
                        const int nt = w.hi - tbase + 1;                            // 0 .. 2*stride-1
                        const int per_w = (nt + T - 1) / T;
                        const int start = tbase + wit * per_w;
                        int m = w.hi - start + 1; if (m > per_w) m = per_w;        // this wave's stretch: [start, start + m)
                        if (m > U.wsz) {
                            const int oa = start + lane, ob = oa + U.wsz;
                            const bool vb = ob < start + m;
                            double pa = U.insd[w.tis0 + w.dir * oa], pb = U.insd[w.tis0 + w.dir * (vb ? ob : oa)];
                            FIG_T0(E);
#ifdef FIG_EMU
                            fig_hot_chain_e2<LDS, 0>(PQ, ncolE, pk, nw2, kt, rs.len, oa + xoff, (vb ? ob : oa) + xoff, pa, pb);
#else
                            fig_hot_chain_e2<LDS, 64>(PQ, ncolE, pk, nw2, kt, rs.len, oa + xoff, 0, pa, pb);
#endif
                            FIG_TICK(E, 9);
                            const double ta = fig_log10(pa), tb = fig_log10(pb);
                            const double wa = fig_exp(0.5 * ta), wb = fig_exp(0.5 * tb);
                            if (ta > best.v) { best.v = ta; best.o = oa; }
                            wrow[oa] = wa;
                            if (vb) { if (tb > best.v) { best.v = tb; best.o = ob; } wrow[ob] = wb; }
                            FIG_TICK(E, 10);
                            nplace += vb ? 2 : 1; nadd += fig_ovl(oa, rs.len, G) + (vb ? fig_ovl(ob, rs.len, G) : 0);
                        } else if (m > 0) {
                            const int oa = start + lane;
                            const bool va = lane < m;
                            const int ca = va ? oa : start;
                            double pa = U.insd[w.tis0 + w.dir * ca];
                            FIG_T0(E);
                            fig_hot_chain_e1<LDS>(PQ, ncolE, pk, kt, rs.len, ca + xoff, pa);
                            const double ta = fig_log10(pa);
                            const double wa = fig_exp(0.5 * ta);
                            if (va) { if (ta > best.v) { best.v = ta; best.o = oa; } wrow[oa] = wa; nplace++; nadd += fig_ovl(oa, rs.len, G); }
                            FIG_TICK(E, 23);
                        }
                    }
                    o = w.hi + 1;
                }
                { FIG_T0(E);
                for (; o <= w.hi; o += stride) {
                    int tis = w.tis0 + w.dir * o;
                    int jstart = clipped ? (-left - o > 0 ? -left - o : 0) : 0;
                    double p = fig_hot_chain_e<LDS>(PQ, Q4, ncolE, pk, nw2, kt, rs.len, jstart, clipped, o + xoff, U.insd[tis]);
                    double t = fig_log10(p);
                    if (t > best.v) { best.v = t; best.o = o; }
                    wrow[o] = fig_exp(0.5 * t);
                    nplace++; nadd += fig_ovl(o, rs.len, G);
                }
                FIG_TICK(E, 24); }
                fl_acc += 4ULL * nplace * (unsigned long long)rs.len + nadd;
                // only the value and "was there any placement" are used downstream (:3680-3688), so a plain maximum will do; the
                // offset of (one of) the maximal placements is kept as a pruning hint for the MLE pass of this placeReads call
                { const double bv = fig_wave_max_dpp(best.o == FIG_NOPOS ? -FIG_DBL_MAX : best.v);
                  const unsigned long long am = fig_ballot(best.o != FIG_NOPOS && best.v == bv);
                  const int ao = am ? fig_lane_read_i32(best.o, fig_ctz64(am)) : FIG_NOPOS;
                  if (lane == 0) { S.wv_v[wave] = bv; S.wv_o[wave] = bv > -FIG_DBL_MAX ? ao : FIG_NOPOS; } }
                if (CPL > 0 && wit == 0) plb[team * 64 + lane] = plv;
            }
        } else {
            // ---- phase A, LDS-tiled class: the table is streamed through the LDS image one column tile at a time.  A placement
            // belongs to the tile that holds its first column (tiles overlap by a read length, so it reads inside the image).
            // Per tile, the chunk's reads are cut into units of 2 x 64 consecutive placements and the units are dealt round-robin
            // over ALL waves, so that the waves reach the tile's barrier together whatever the reads' windows look like.
            const int nrd = nU - c0 < nteams ? nU - c0 : nteams;
            for (int t = 0; t < nrd; t++) {
                const FigReadS rs = fig_read_scalars(U, ub + c0 + t);
                const FigWin w = fig_window_u(U, rs.pos, rs.len, gapoffset);
                if (tid == 0) { S.tm_lo[t] = w.lo; S.tm_hi[t] = w.hi; S.tm_len[t] = rs.len; S.tm_tis0[t] = w.tis0; S.tm_dir[t] = w.dir;
                                S.tm_aux[t] = rs.rev | (rs.hasN << 1); S.tm_woff[t] = rs.woff; }
                if (lane == 0) { S.wv_v[t * U.nw + wave] = -FIG_DBL_MAX; S.wv_o[t * U.nw + wave] = FIG_NOPOS; }
                double *wrow = W + (long long)t * Wcap + (U.L - 1);
                for (int i = -(U.L - 1) + tid; i < G; i += U.nt) if (i < w.lo || i > w.hi) wrow[i] = 0.0;
                if (CPL > 0 && wave == t) {                     // stage {counts, position lists} for the column pass (t < nteams <= nw)
                    fig_cu32p pk = (fig_cu32p)(U.packed + rs.woff);
                    const int ndw = 2 + ((rs.len + 3) >> 2) + 4;
                    plb[t * 64 + lane] = lane < ndw ? pk[((rs.len + 15) >> 4) + ((rs.len + 31) >> 5) + lane] : 0u;
                }
            }
            for (int tl = 0; tl < ntl; tl++) {
                const int x0 = tl * tstep;
                FIG_SYNC();                                   // the chains of the previous tile are done with the image
                fig_tile_copy(E, x0, pstride);
                FIG_SYNC();
                const int xs = xoff - x0;                     // image column of placement offset 0
                int ubase = 0;
                for (int t = 0; t < nrd; t++) {
                    // the read's scalars come from the copy staged in LDS (a scalar-load round trip per read and tile otherwise)
                    FigReadS rs; FigWin w;
                    w.lo = fig_u(S.tm_lo[t]); w.hi = fig_u(S.tm_hi[t]); w.tis0 = fig_u(S.tm_tis0[t]); w.dir = fig_u(S.tm_dir[t]);
                    rs.len = fig_u(S.tm_len[t]); { const int a = fig_u(S.tm_aux[t]); rs.rev = a & 1; rs.hasN = (a >> 1) & 1; }
                    rs.pos = 0; rs.woff = fig_u64(S.tm_woff[t]);
                    int lo_t = w.lo, hi_t = w.hi;
                    if (tl > 0 && x0 - xoff > lo_t) lo_t = x0 - xoff;
                    if (tl + 1 < ntl && x0 + tstep - 1 - xoff < hi_t) hi_t = x0 + tstep - 1 - xoff;
                    if (hi_t < lo_t) continue;
                    const int usz = 2 * U.wsz;
                    const int nun = (hi_t - lo_t + usz) / usz;
                    int k = (wave - ubase % U.nw + U.nw) % U.nw;
                    ubase += nun;
                    if (k >= nun) continue;
                    fig_cu32p pk = (fig_cu32p)(U.packed + rs.woff);
                    const int nw2 = (rs.len + 15) >> 4;
                    fig_cdp kt = (fig_cdp)(rs.rev ? U.kt_rev + 2 * (U.L - rs.len) : U.kt_fwd);
                    double *wrow = W + (long long)t * Wcap + (U.L - 1);
                    const bool generic = rs.hasN != 0 || clipped;
                    FigBest best; best.v = -FIG_DBL_MAX; best.o = FIG_NOPOS;
                    unsigned long long nplace = 0, nadd = 0;
                    for (; k < nun; k += U.nw) {
                        const int base = lo_t + k * usz;
                        const int oa = base + lane, ob = oa + U.wsz;
                        const bool va = oa <= hi_t, vb = ob <= hi_t;
                        const int ca = va ? oa : base, cb = vb ? ob : base;
                        double pa = U.insd[w.tis0 + w.dir * ca], pb = U.insd[w.tis0 + w.dir * cb];
                        FIG_T0(E);
                        if (!generic) {
#ifdef FIG_EMU
                            fig_hot_chain_e2<LDS, 0>(PQ, pstride, pk, nw2, kt, rs.len, ca + xs, cb + xs, pa, pb);
#else
                            fig_hot_chain_e2<LDS, 64>(PQ, pstride, pk, nw2, kt, rs.len, base + lane + xs, 0, pa, pb);
#endif
                        } else {
                            if (va) pa = fig_hot_chain_e<LDS>(PQ, Q4, pstride, pk, nw2, kt, rs.len, clipped ? (-left - oa > 0 ? -left - oa : 0) : 0, clipped, oa + xs, pa);
                            if (vb) pb = fig_hot_chain_e<LDS>(PQ, Q4, pstride, pk, nw2, kt, rs.len, clipped ? (-left - ob > 0 ? -left - ob : 0) : 0, clipped, ob + xs, pb);
                        }
                        FIG_TICK(E, 9);
                        const double ta = fig_log10(pa), tb = fig_log10(pb);
                        const double wa = fig_exp(0.5 * ta), wb = fig_exp(0.5 * tb);
                        if (va) { if (ta > best.v) { best.v = ta; best.o = oa; } wrow[oa] = wa; nplace++; nadd += fig_ovl(oa, rs.len, G); }
                        if (vb) { if (tb > best.v) { best.v = tb; best.o = ob; } wrow[ob] = wb; nplace++; nadd += fig_ovl(ob, rs.len, G); }
                        FIG_TICK(E, 10);
                    }
                    fl_acc += 4ULL * nplace * (unsigned long long)rs.len + nadd;
                    // fold this wave's arg-max of read t into its own slot (no other wave writes it)
                    { const double bv = fig_wave_max_dpp(best.o == FIG_NOPOS ? -FIG_DBL_MAX : best.v);
                      const unsigned long long am = fig_ballot(best.o != FIG_NOPOS && best.v == bv);
                      if (am) {
                          const int ao = fig_lane_read_i32(best.o, fig_ctz64(am));
                          const int sl = t * U.nw + wave;
                          if (lane == 0 && (S.wv_o[sl] == FIG_NOPOS || bv > S.wv_v[sl])) { S.wv_v[sl] = bv; S.wv_o[sl] = ao; }
                      } }
                }
            }
        }
        FIG_TICK(E, 11);
        FIG_SYNC();
        FIG_TICK(E, 0);
        // ---- per-read bookkeeping (:3680-3688)
        if (tid < nteams && c0 + tid < nU) {
            FigBest b; b.v = -FIG_DBL_MAX; b.o = FIG_NOPOS;
            for (int k = 0; k < Tm; k++) { FigBest y; y.v = S.wv_v[tid * Tm + k]; y.o = S.wv_o[tid * Tm + k]; b = fig_best_merge(b, y); }
            if (b.o != FIG_NOPOS) E.scr.maxlv[c0 + tid] = b.v;
            else { E.scr.maxlv[c0 + tid] = 0; fig_atomic_add_i32(&S.invalid_count, 1); }
            E.scr.hint_e[c0 + tid] = b.o;
        }
        // ---- phase B: lanes = gap columns; reads of the chunk in order.  For one read the additions into the
        // five per-base accumulators of a column are independent chains; each chain walks that base's positions
        // in descending j (= ascending placement), which is the reference's order (:3603-3611).
        for (int t = 0; t < nteams && c0 + t < nU; t++) {
#ifndef FIG_EMU
            if (U.nw == 8) { if ((t + (wave >> 2)) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
#endif
            const int lo = fig_u(S.tm_lo[t]), hi = fig_u(S.tm_hi[t]), len = fig_u(S.tm_len[t]);
            if (hi < lo) continue;
            long long woff = fig_u64(U.u_woff[ub + c0 + t]);
            fig_cu32p pk = (fig_cu32p)(U.packed + woff);
            const int nw2 = (len + 15) >> 4, nwm = (len + 31) >> 5;
            const double *wrow = W + (long long)t * Wcap + (U.L - 1);
            if (CPL == 0) {
                for (int x = tid; x < G; x += U.nt) {
                    double a[5];
                    for (int b = 0; b < 5; b++) a[b] = E.scr.cnt[b * cg + x];
                    for (int jj = len - 1; jj >= 0; jj--) a[fig_sbase(pk, nw2, jj)] += wrow[x - jj];
                    for (int b = 0; b < 5; b++) E.scr.cnt[b * cg + x] = a[b];
                }
                continue;
            }
#ifndef FIG_EMU
            if (CPL > 0) {
                const uint32_t plv = plb[t * 64 + lane];
                const uint32_t cw0 = (uint32_t)__builtin_amdgcn_readlane((int)plv, 0), cw1 = (uint32_t)__builtin_amdgcn_readlane((int)plv, 1);
                const int nA = (int)(cw0 & 255), nC = (int)((cw0 >> 8) & 255), nG = (int)((cw0 >> 16) & 255), nT = (int)(cw0 >> 24);
                const int n4 = (int)(cw1 & 255);
                const int dC = 2 + ((nA + 3) >> 2), dG = dC + ((nC + 3) >> 2), dT = dG + ((nG + 3) >> 2), d4 = dT + ((nT + 3) >> 2);
                const int n = pb_c == 0 ? nA : pb_c == 1 ? nC : pb_c == 2 ? nG : nT;
                const int d0 = pb_c == 0 ? 2 : pb_c == 1 ? dC : pb_c == 2 ? dG : dT;
                // tile i of this wave reads wl[i * 64 - j]: constant offsets from one per-position address
                const double *wl = wrow + pb_x0;
                int k4 = 0;
                for (; k4 + 1 <= (n >> 2); k4++) {
                    const uint32_t s4 = (uint32_t)__builtin_amdgcn_readlane((int)plv, d0 + k4);
                    const double *p0 = wl - (int)(s4 & 255), *p1 = wl - (int)((s4 >> 8) & 255), *p2 = wl - (int)((s4 >> 16) & 255), *p3 = wl - (int)(s4 >> 24);
                    if (CPL <= 4) {
                        double w0[CPL > 0 ? CPL : 1], w1[CPL > 0 ? CPL : 1], w2[CPL > 0 ? CPL : 1], w3[CPL > 0 ? CPL : 1];
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
                        double w0[CPL > 0 ? CPL : 1], w1[CPL > 0 ? CPL : 1];
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
                if (n & 3) {
                    uint32_t s4 = (uint32_t)__builtin_amdgcn_readlane((int)plv, d0 + k4);
                    for (int q = 0; q < (n & 3); q++, s4 >>= 8) {
                        const double *p0 = wl - (int)(s4 & 255);
#pragma unroll
                        for (int i = 0; i < CPL; i++) acc[i] += FIG_LDV(p0 + i * 64);
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
            }
#endif
        }
#ifndef FIG_EMU
        if (U.nw == 8) __builtin_amdgcn_s_setprio(0);
#endif
        FIG_TICK(E, 12);
        FIG_SYNC();
        FIG_TICK(E, 1);
    }
    if (CPL > 0) {
#pragma unroll
        for (int i = 0; i < CPL; i++) {
            const int x = pb_x0 + i * U.wsz;
            if (x < G) E.scr.cnt[pb_c * cg + x] = acc[i];
        }
    }
    E.flops += fl_acc;
    FIG_SYNC();
}

#include "fig_engine_shared.h"

template <bool LDS>
FIG_D void fig_hot_estep_dispatch(FigEng &E, int gapoffset) {
#ifdef FIG_EMU
    if (LDS && E.tiles > 0) fig_hot_estep<LDS, 0, true>(E, gapoffset);
    else fig_hot_estep<LDS, 0, false>(E, gapoffset);
#else
    const int ntiles = (E.S->G + E.wsz - 1) / E.wsz;
    const int nsub = E.nw >> 2;                          // waves per base (workgroups are 4 or 8 waves)
    const int cpl = nsub > 0 ? (ntiles + nsub - 1) / nsub : 99;      // tiles per wave
    if (LDS && E.tiles > 0) {                            // LDS-tiled class: wide gaps only, but a candidate can be much shorter than the class
        if (cpl <= 8) fig_hot_estep<LDS, 8, true>(E, gapoffset);
        else if (cpl <= 12) fig_hot_estep<LDS, 12, true>(E, gapoffset);
        else if (cpl <= 16) fig_hot_estep<LDS, 16, true>(E, gapoffset);
        else if (cpl <= 24) fig_hot_estep<LDS, 24, true>(E, gapoffset);          // gaps of 2049-3072 bp
        else if (cpl <= 32) fig_hot_estep<LDS, 32, true>(E, gapoffset);          // ... -4096 bp
        else fig_hot_estep<LDS, 0, true>(E, gapoffset);
        return;
    }
    if (LDS && fig_sh_applies(E, cpl)) {
        switch (cpl) {
            case 1: fig_hot_estep_sh<1>(E, gapoffset); break;
            case 2: fig_hot_estep_sh<2>(E, gapoffset); break;
            case 3: fig_hot_estep_sh<3>(E, gapoffset); break;
            case 4: fig_hot_estep_sh<4>(E, gapoffset); break;
            case 5: fig_hot_estep_sh<5>(E, gapoffset); break;
            case 6: fig_hot_estep_sh<6>(E, gapoffset); break;
            case 7: fig_hot_estep_sh<7>(E, gapoffset); break;
            default: fig_hot_estep_sh<8>(E, gapoffset); break;
        }
        return;
    }
    switch (cpl) {
        case 1: fig_hot_estep<LDS, 1, false>(E, gapoffset); break;
        case 2: fig_hot_estep<LDS, 2, false>(E, gapoffset); break;
        case 3: fig_hot_estep<LDS, 3, false>(E, gapoffset); break;
        case 4: fig_hot_estep<LDS, 4, false>(E, gapoffset); break;
        case 5: fig_hot_estep<LDS, 5, false>(E, gapoffset); break;
        case 6: fig_hot_estep<LDS, 6, false>(E, gapoffset); break;
        case 7: fig_hot_estep<LDS, 7, false>(E, gapoffset); break;
        case 8: fig_hot_estep<LDS, 8, false>(E, gapoffset); break;
        case 9: case 10: fig_hot_estep<LDS, 10, false>(E, gapoffset); break;
        case 11: case 12: fig_hot_estep<LDS, 12, false>(E, gapoffset); break;
        case 13: case 14: case 15: case 16: fig_hot_estep<LDS, 16, false>(E, gapoffset); break;
        default: fig_hot_estep<LDS, 0, false>(E, gapoffset); break;
    }
#endif
}

// Full MLE product of ONE placement, by the whole wave: the lanes compute the read's len factors side by side
// (factor j = m3[j] on a match, e[j]*T[from][to] otherwise -- the same two operations as the chain), park them in
// fbuf, and then every lane multiplies them up in the reference's order j = 0, 1, ...; so the result is bit-identical
// to the lane-per-placement chain but costs ~len dependent multiplies instead of a whole 128-placement round.
// fbuf: >= len doubles private to the wave (LDS ops of one wave execute in order, so no barrier is needed).
// What a lane needs of the read for its factors j = lane, lane + wsz, ...: loaded once per read with vector loads
// (their latency hides behind the first chain blocks), then reused by every serial evaluation of that read.
struct FigSerLane { int b[4]; double m3[4], e[4]; };
FIG_D void fig_mle_serial_prep(FigSerLane &L, fig_cu32p pk, fig_cdp mt, int len, int lane, int wsz) {
#ifndef FIG_EMU
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int j = lane + k * wsz;
        const int jj = j < len ? j : 0;
        L.b[k] = (int)((pk[jj >> 4] >> ((jj & 15) * 2)) & 3);
        L.m3[k] = mt[2 * jj]; L.e[k] = mt[2 * jj + 1];
    }
#else
    (void)L; (void)pk; (void)mt; (void)len; (void)lane; (void)wsz;
#endif
}
template <bool LDS>
FIG_D double fig_mle_serial(const FigSerLane &L, const double *C, int ncolE, fig_cu32p pk, fig_cdp mt, int len, int xe0, double *fbuf, int lane, int wsz) {
#ifndef FIG_EMU
    (void)pk; (void)mt;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int j = lane + k * wsz;
        if (j < len) {
            const double c = C[L.b[k] * ncolE + xe0 + j];
            const double f = L.e[k] * c;
            fbuf[j] = c < 0 ? L.m3[k] : f;
        }
    }
#else
    (void)L;
    for (int j = lane; j < len; j += wsz) {
        const int b = (int)((pk[j >> 4] >> ((j & 15) * 2)) & 3);
        const double m3 = mt[2 * j], e = mt[2 * j + 1];
        const double c = C[b * ncolE + xe0 + j];
        const double f = e * c;
        fbuf[j] = c < 0 ? m3 : f;
    }
#endif
    double q = 1;
    int j = 0;
    for (; j + 16 <= len; j += 16) {
        double f[16];
#pragma unroll
        for (int k = 0; k < 16; k++) f[k] = fbuf[j + k];
#pragma unroll
        for (int k = 0; k < 16; k++) q *= f[k];
    }
    for (; j < len; j++) q *= fbuf[j];
    return q;
}

// ---------------------------------------------------------------------------------------
// MLE pass over all unmapped reads.  mode 0: placeReads (Figbird.cpp:3732-3914); mode 1: finalize
// (:5018-5192).  `gl` = length of the gap string the products are taken against (== S.G except inside
// finalize, where the string was cut with the previous window), wl/wr = that window's flank extents.
// Results: E.scr.accf[r], per-read placement in frp (mode 0) / fin (mode 1), integer pile-up in nci,
// valid_count / flags; the ordered likelihood sum is done afterwards by the caller.
template <bool LDS>
FIG_D void fig_hot_mle(FigEng &E, int gapoffset, int mode, int gl, int wl, int wr) {
    FigState &S = *E.S;
    const FigHotU U = fig_hot_uniforms(E);
    const int G = U.G, left = U.left, nU = U.nU, cg = U.cg, G0 = U.G0;
    const long long ub = U.ub;
    const int ncolE = U.ncolE, xoff = U.xoff;
    // one read per wave: the MLE pass keeps no weight rows, and a wave that owns the whole read prunes every round
    // but the hinted one with its own running maximum
    const int nteams = U.nw;
    const int T = 1;
    const int wave = fig_u(E.wave), lane = E.lane;
    const int team = wave, wit = 0;
    const int ncl = mode == 0 ? S.ncols : (gl > cg ? cg : gl);
    // integer pile-up of the accepted reads: in the idle weight rows (LDS atomics) when they have room behind the
    // per-wave factor buffers, else straight in the scratch slab; copied out after the pass
    int nrows = U.nteams; if (nrows > U.nw) nrows = U.nw;
    const bool use_serial = (long long)nrows * U.Wcap >= (long long)U.nw * FIG_MLE_FB;
    const long long fb_doubles = use_serial ? (long long)U.nw * FIG_MLE_FB : 0;
    const bool nci_lds = LDS && (long long)nrows * U.Wcap - fb_doubles >= (5LL * ncl + 1) / 2 + 1;
    int *nl = nci_lds ? (int *)(fig_w_ptr<LDS>(E) + fb_doubles) : E.scr.nci;
    const int nst = nci_lds ? ncl : cg;             // row stride of the pile-up
    for (int x = E.tid; x < ncl; x += E.nt) for (int j = 0; j < 5; j++) nl[j * nst + x] = 0;
    if (E.tid == 0) S.mle_next = 0;
    // Mismatch filter: the consensus codes of the extended columns packed 16 per word (kc: 2-bit code, kn: bit 2i set
    // when column i is N/outside), so that counting the mismatches of a whole read against a placement is ~len/16 xor +
    // popcount steps.  A match factor is <= 1 and a mismatch factor <= fmm_up, so a placement with m mismatches has a
    // product <= fmm_up^m: once a running maximum is known, placements with too many mismatches are out without a
    // single FP64 operation, and the few others are evaluated one by one (fig_mle_serial).
    const int kwords = (ncolE + 15) / 16 + 16;       // + a read's worth of words past the last column (read as N)
    const long long nci_doubles = nci_lds ? (5LL * ncl + 1) / 2 + 1 : 0;
    const double fmm = fig_uptr(E.M)->fmm_up;
    const bool mono = fmm <= 1.0;                     // factors in [0,1]: partial products never grow (else: no pruning at all)
    const double lfmm = fig_log(fmm < 1.0 ? fmm : 0.5);
    const bool use_kf = LDS && fmm < 0.5 && (long long)nrows * U.Wcap - fb_doubles - nci_doubles >= kwords + 1;
    uint32_t *kc = (uint32_t *)(fig_w_ptr<LDS>(E) + fb_doubles + nci_doubles), *kn = kc + kwords;
    if (use_kf) {
        int ncb = gl + 2 * xoff; if (ncb > ncolE) ncb = ncolE;
        const fig_gcu8p fl = (fig_gcu8p)fig_uptr(E.B)->flank + fig_u64(fig_uptr(E.g)->flankOff);
        const fig_lcu8p gsl = (fig_lcu8p)E.gs;
        for (int wd = E.tid; wd < kwords; wd += E.nt) {
            uint32_t c2 = 0, n2 = 0;
            for (int q = 0; q < 16; q++) {
                const int i = wd * 16 + q;
                const int from = i < ncb ? fig_from_code_u(fl, gsl, i - xoff, gl, wl, wr) : 4;
                if (from >= 0 && from < 4) c2 |= (uint32_t)from << (2 * q); else n2 |= 1u << (2 * q);
            }
            kc[wd] = c2; kn[wd] = n2;
        }
    }
    // 32-column windows of the packed consensus, one 16-byte record per column {codes lo, codes hi, N lo, N hi}, in the part
    // of the table area the MLE table leaves free (C uses 5 of the 9 doubles per column): the sweep below then needs ONE
    // ds_read_b128 per placement instead of six word reads and four funnel shifts
    const bool use_cw = LDS && use_kf;
    FigU4 *cwn = (FigU4 *)(fig_lds + fig_u(E.off_pq) + 5LL * ncolE);
    if (use_cw) {
        FIG_SYNC();                                   // kc / kn complete
        for (int i = E.tid; i < ncolE; i += E.nt) {
            const int wi = i >> 4, sh = (i & 15) * 2;
            const uint32_t a0 = kc[wi], a1 = kc[wi + 1], a2 = kc[wi + 2], b0 = kn[wi], b1 = kn[wi + 1], b2 = kn[wi + 2];
            FigU4 v;
            v.x = sh ? (a0 >> sh) | (a1 << (32 - sh)) : a0; v.y = sh ? (a1 >> sh) | (a2 << (32 - sh)) : a1;
            v.z = sh ? (b0 >> sh) | (b1 << (32 - sh)) : b0; v.w = sh ? (b1 >> sh) | (b2 << (32 - sh)) : b1;
            cwn[i] = v;
        }
    }
    fig_build_mle_table(E, gl, wl, wr);             // ends with a barrier
    const double *C = fig_c_ptr<LDS>(E);
    const double init = mode == 0 ? -FIG_DBL_MAX : 0.0;
    unsigned long long fl_acc = 0;                   // lane-local flop count, added to E.flops once
    unsigned long long ex_acc = 0;                   // wave-uniform: multiplies into placement products actually executed
    // per-wave factor buffer of fig_mle_serial: the weight rows are idle during the MLE pass
    double *fbuf = fig_w_ptr<LDS>(E) + (use_serial ? wave * FIG_MLE_FB : 0);
    FIG_T0(E);
    (void)nteams; (void)team;
    // ---- Lane-per-read path.  For reads of full length without N (nearly all of them) the whole search runs with ONE READ
    // PER LANE, 64 reads side by side, so that nothing in it waits on a wave-uniform dependency chain:
    //   (1) hints: the MLE product at the previous pass's best placement and at the E-step's arg-max (same factors, same
    //       order as fig_mle_serial / the chains: same bits); the better one is the read's running maximum `bound`;
    //   (2) sweep: every placement of the window is judged by the mismatches of its first 32 bases against the packed
    //       consensus (one 16-byte record per placement); a placement with m mismatches has a product <= fmm^m, so only
    //       those with m < mcut(bound) can reach the maximum; the few that pass get the whole-read count, and those that
    //       pass that too (at most two per read, else the read is left to the wave-per-read loop below) are
    //   (3) evaluated, again lane-parallel, and merged "first maximum wins";
    //   (4) the accept test, flags and integer pile-up of these reads follow after the wave-per-read loop (same order-free
    //       integer / flag updates as there).
    // Value and arg-max are exactly those of the full scan: every placement that could reach or tie the hint's product
    // passes both filters.
    unsigned char *mdone = E.scr.mdone;
    const bool lane_path = use_cw && mono;
    {
        fig_cdp mtf = (fig_cdp)U.mt_fwd, mtr = (fig_cdp)U.mt_rev;
        const int Lfull = U.L;
        const int nw2f = (Lfull + 15) >> 4;
        for (int r0 = 0; r0 < nU; r0 += U.nt) {
            const int r = r0 + E.tid;
            bool part = false;
            int h1 = FIG_NOPOS, h2 = FIG_NOPOS, rev = 0;
            fig_gcu32p pkl = (fig_gcu32p)U.packed;          // global view: a generic pointer's FLAT loads would hold up the LDS waits below
            FigWin w; w.lo = 0; w.hi = -1; w.tis0 = 0; w.dir = 1;
            if (r < nU) {
                const int len = U.u_len[ub + r], aux = U.u_aux[ub + r];
                rev = aux & 1;
                mdone[r] = 0;
                E.scr.hval[r] = -1.0;
                if (len == Lfull && !((aux >> 1) & 1)) {
                    w = fig_window_u(U, U.u_pos[ub + r], len, gapoffset);
                    h1 = E.scr.hint[r]; h2 = E.scr.hint_e[r];
                    if (h1 != FIG_NOPOS && (h1 < w.lo || h1 > w.hi)) h1 = FIG_NOPOS;
                    if (h2 != FIG_NOPOS && (h2 < w.lo || h2 > w.hi || h2 == h1)) h2 = FIG_NOPOS;
                    if (h1 == FIG_NOPOS) { h1 = h2; h2 = FIG_NOPOS; }
                    part = h1 != FIG_NOPOS;
                    pkl = (fig_gcu32p)U.packed + U.u_woff[ub + r];
                }
            }
            if (!fig_wave_any(part)) continue;
            // (1) + (3): up to two candidate placements per lane, one dependent multiply chain each
            FigBest a; a.v = init; a.o = FIG_NOPOS;
            int c1 = h1, c2 = h2;
            bool sw = false, ovf = false; int mcut = 0;
            for (int pass = 0; pass < 2; pass++) {
                const bool ev = part && c1 != FIG_NOPOS;
                if (fig_wave_any(ev)) {
                    const bool two = fig_wave_any(ev && c2 != FIG_NOPOS);
                    const int xa = (ev ? c1 : 0) + xoff, xb = (ev && c2 != FIG_NOPOS ? c2 : (ev ? c1 : 0)) + xoff;
                    double qa = 1, qb = 1;
                    uint32_t wd_nx = ev ? pkl[0] : 0u;
                    for (int wi = 0; wi < nw2f; wi++) {
                        const uint32_t wd = wd_nx;                          // fetched a word (16 steps) ahead
                        if (wi + 1 < nw2f) wd_nx = ev ? pkl[wi + 1] : 0u;
                        const int j0 = wi * 16;
                        int nb = Lfull - j0; if (nb > 16) nb = 16;
                        if (nb == 16) {
                            // a full word of the read: eight columns' entries fetched together, then multiplied in in order
                            // (same operations, same order: the loop below with its loads hoisted)
#pragma unroll
                            for (int h = 0; h < 2; h++) {
                                double ca8[8], cb8[8];
#pragma unroll
                                for (int u = 0; u < 8; u++) {
                                    const int jj = 8 * h + u, j = j0 + jj;
                                    const int b = (int)((wd >> (2 * jj)) & 3);
                                    ca8[u] = C[b * ncolE + xa + j];
                                    cb8[u] = two ? C[b * ncolE + xb + j] : 0.0;
                                }
#pragma unroll
                                for (int u = 0; u < 8; u++) {
                                    const int j = j0 + 8 * h + u;
                                    const double m3 = rev ? mtr[2 * j] : mtf[2 * j], e = rev ? mtr[2 * j + 1] : mtf[2 * j + 1];
                                    const double fa = e * ca8[u];
                                    qa *= (ca8[u] < 0 ? m3 : fa);
                                    if (two) { const double fb = e * cb8[u]; qb *= (cb8[u] < 0 ? m3 : fb); }
                                }
                            }
                            continue;
                        }
                        for (int jj = 0; jj < nb; jj++) {
                            const int j = j0 + jj;
                            const int b = (int)((wd >> (2 * jj)) & 3);
                            const double m3 = rev ? mtr[2 * j] : mtf[2 * j], e = rev ? mtr[2 * j + 1] : mtf[2 * j + 1];
                            const double ca = C[b * ncolE + xa + j];
                            const double fa = e * ca;
                            qa *= (ca < 0 ? m3 : fa);
                            if (two) {
                                const double cb = C[b * ncolE + xb + j];
                                const double fb = e * cb;
                                qb *= (cb < 0 ? m3 : fb);
                            }
                        }
                    }
                    ex_acc += (unsigned long long)Lfull * (unsigned long long)(fig_popc64(fig_ballot(ev)) + fig_popc64(fig_ballot(ev && c2 != FIG_NOPOS)));
                    if (ev) {
                        if (qa > init) { FigBest y; y.v = qa; y.o = c1; a = fig_best_merge(a, y); }
                        if (c2 != FIG_NOPOS && qb > init) { FigBest y; y.v = qb; y.o = c2; a = fig_best_merge(a, y); }
                    }
                }
                if (pass == 1) break;
                // (2) the sweep, for lanes whose hint gives a sharp cut
                sw = lane_path && part && a.o != FIG_NOPOS && a.v > 1e-290 && Lfull >= 32;
                if (sw) { mcut = (int)(fig_log(a.v) / lfmm) + 2; sw = mcut <= 18; }
                c1 = c2 = FIG_NOPOS;
                if (!fig_wave_any(sw)) break;
                const uint32_t rw0 = sw ? pkl[0] : 0u, rw1 = sw ? pkl[1] : 0u;
                const int Wn = sw ? w.hi - w.lo + 1 : 0;
                const FigU4 *cwx = cwn + xoff;
                // four placements per trip: their records are fetched together (one LDS round trip per four tests instead of one per
                // test; a lane past its window re-reads its last record and fails the `i < Wn` test), judged in placement order
                for (int i0 = 0; fig_wave_any(i0 < Wn); i0 += 4) {
                  FigU4 rec4[4];
#pragma unroll
                  for (int u = 0; u < 4; u++) { const int iu = i0 + u < Wn ? i0 + u : (Wn > 0 ? Wn - 1 : 0); rec4[u] = cwx[w.lo + iu]; }
#pragma unroll
                  for (int u = 0; u < 4; u++) {
                    const int i = i0 + u;
                    const int o = w.lo + i;
                    const FigU4 rec = rec4[u];
                    const uint32_t d0 = rec.x ^ rw0, d1 = rec.y ^ rw1;
                    const int mm = __builtin_popcount(((d0 | (d0 >> 1)) | rec.z) & 0x55555555u) + __builtin_popcount(((d1 | (d1 >> 1)) | rec.w) & 0x55555555u);
                    if (i < Wn && mm < mcut && o != h1 && o != h2) {
                        int mmf = 0;                                  // whole-read count, 32 bases per record
                        for (int k = 0; 32 * k < Lfull; k++) {
                            const FigU4 rk = cwx[o + 32 * k];
                            const int nbk = Lfull - 32 * k;
                            const uint32_t mlo = nbk >= 16 ? 0x55555555u : (0x55555555u >> (2 * (16 - nbk)));
                            const uint32_t mhi = nbk >= 32 ? 0x55555555u : (nbk > 16 ? (0x55555555u >> (2 * (32 - nbk))) : 0u);
                            const uint32_t e0 = rk.x ^ pkl[2 * k], e1 = rk.y ^ pkl[2 * k + 1];
                            mmf += __builtin_popcount(((e0 | (e0 >> 1)) | rk.z) & mlo) + __builtin_popcount(((e1 | (e1 >> 1)) | rk.w) & mhi);
                        }
                        if (mmf < mcut) { if (c1 == FIG_NOPOS) c1 = o; else if (c2 == FIG_NOPOS) c2 = o; else ovf = true; }
                    }
                  }
                }
            }
            if (part) {
                E.scr.hint[r] = a.o;
                if (sw && !ovf) { E.scr.hval[r] = a.v; mdone[r] = 1; fl_acc += (unsigned long long)(w.hi - w.lo + 1) * (unsigned long long)Lfull; }
                else if (!sw && a.o != FIG_NOPOS) E.scr.hval[r] = a.v;     // the wave-per-read loop starts from this maximum
                // (sw && ovf: too many survivors for two slots; hval stays -1 and the loop below redoes the read in full)
            }
        }
        FIG_SYNC();
    }
    FIG_TICK(E, 18);
    while (true) {
        // waves take reads dynamically (reads differ a lot in cost once pruning works), so the pass ends evenly
        int r = 0;
        if (lane == 0) r = fig_atomic_fetch_add_i32(&S.mle_next, 1);
        r = fig_u(r);
        if (r >= nU) break;
        if (mdone[r]) continue;                          // finished by the lane-per-read path; accepted after this loop
        const bool active = true;
        FigReadS rs; rs.len = 0; rs.rev = 0; rs.hasN = 0; rs.pos = 0; rs.woff = 0;
        FigWin w; w.lo = 0; w.hi = -1; w.tis0 = 0; w.dir = 1;
        if (active) {
            rs = fig_read_scalars(U, ub + r);
            w = fig_window_u(U, rs.pos, rs.len, gapoffset);
            fig_cu32p pk = (fig_cu32p)(U.packed + rs.woff);
            int nw2 = (rs.len + 15) >> 4;
            fig_cdp mt = (fig_cdp)(rs.rev ? U.mt_rev + 2 * (U.L - rs.len) : U.mt_fwd);
            FigBest best; best.v = init; best.o = FIG_NOPOS;
            unsigned long long nplace = 0;
            const bool hasN = rs.hasN != 0;
            const int stride = T * U.wsz;
            int o = w.lo + wit * U.wsz + lane;
            if (!hasN) {
                // pair-rounds of this wave: round rr covers o = obase + 2*rr*stride (and + stride).  Start with the round
                // that holds the previous pass's best placement: it almost always yields the final maximum, after which
                // every other round is abandoned after its first 8-base block.
                const int obase = w.lo + wit * U.wsz;               // lane 0 of round 0
                const int span = w.hi - obase;                      // < 0: nothing for this wave
                const int nrounds = span < 0 ? 0 : span / (2 * stride) + 1;
                // Survivor scheme.  Every factor is <= 1, so a placement's product over a prefix of the read bounds its full
                // product from above.  With a running maximum `bound` in hand (first from the hinted placement, evaluated on
                // its own by fig_mle_serial) a round runs the pruned pair chain only until few lanes are still above the
                // bound; those are then finished one by one, again serially.  Nothing else can be the maximum (or tie it),
                // so value and arg-max are exactly those of the full scan.
                double bound = init;
                int mcut_cur = 0x7fffffff; double mcut_for = init;   // mismatch cut of the filter, recomputed when bound moves
                FigBest ub; ub.v = init; ub.o = FIG_NOPOS;         // wave-uniform best of the serial evaluations
                FigSerLane SL;
                if (use_serial) fig_mle_serial_prep(SL, pk, mt, rs.len, lane, U.wsz);
                FIG_TICK(E, 14);
                const int h = fig_u(E.scr.hint[r]);
                int r0 = 0;
                if (h != FIG_NOPOS && h >= obase && h <= w.hi) {
                    r0 = (h - obase) / (2 * stride); if (r0 >= nrounds) r0 = 0;
                    if (use_serial) {
                        double v = fig_bcast_d(E.scr.hval[r]);          // evaluated lane-per-read above (full-length reads)
                        if (v < 0) { ex_acc += (unsigned long long)rs.len; v = fig_mle_serial<LDS>(SL, C, ncolE, pk, mt, rs.len, h + xoff, fbuf, lane, U.wsz); }
                        if (v > init) { ub.v = v; ub.o = h; bound = v; }
                    }
                }
                FIG_TICK(E, 15);
                // Fast path (a usable maximum from the hint): one sweep over all placements, one per lane and step, judging
                // each by the mismatches of its first 32 bases; the few that pass get the whole-read count (lanes = words
                // of the read) and, if that passes too, the one-by-one evaluation.
                bool swept = false;
                if (use_kf && use_serial && bound > 1e-290 && rs.len >= 32) { mcut_cur = (int)(fig_log(bound) / lfmm) + 2; mcut_for = bound; }
                // (only with a sharp cut: 32 random bases show ~24 mismatches, so a cut above ~18 would let too many through)
                if (use_kf && use_serial && bound > 1e-290 && rs.len >= 32 && mcut_cur <= 18) {
                    swept = true;
                    const uint32_t rw0 = pk[0], rw1 = pk[1];
                    const int nwr = (rs.len + 15) >> 4;
                    const int lwq = lane < nwr ? lane : nwr - 1;                 // the read word this lane owns in the full count
                    const uint32_t rwl = pk[lwq];
                    const int reml = rs.len - lwq * 16;
                    const uint32_t lml = lane < nwr ? (reml >= 16 ? 0x55555555u : (0x55555555u >> (2 * (16 - reml)))) : 0u;
                    // four 64-placement groups per trip: their records are fetched together (one LDS latency per trip) and
                    // nearly every trip ends at the combined ballot
                    for (int ob0 = w.lo; ob0 <= w.hi; ob0 += 4 * U.wsz) {
                        unsigned long long cm4[4];
                        FigU4 rec4[4]; bool ok4[4];
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            const int o1 = ob0 + k * U.wsz + lane;
                            ok4[k] = o1 <= w.hi;
                            const int xe = (ok4[k] ? o1 : w.hi) + xoff;
                            if (use_cw) rec4[k] = cwn[xe];
                            else {
                                const int wi = xe >> 4, sh = (xe & 15) * 2;
                                const uint32_t c0 = kc[wi], c1 = kc[wi + 1], c2 = kc[wi + 2], n0 = kn[wi], n1 = kn[wi + 1], n2 = kn[wi + 2];
                                rec4[k].x = sh ? (c0 >> sh) | (c1 << (32 - sh)) : c0; rec4[k].z = sh ? (n0 >> sh) | (n1 << (32 - sh)) : n0;
                                rec4[k].y = sh ? (c1 >> sh) | (c2 << (32 - sh)) : c1; rec4[k].w = sh ? (n1 >> sh) | (n2 << (32 - sh)) : n1;
                            }
                        }
                        if (bound != mcut_for) { mcut_cur = (int)(fig_log(bound) / lfmm) + 2; mcut_for = bound; }
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            if (ok4[k]) nplace++;
                            const uint32_t d0 = rec4[k].x ^ rw0, d1 = rec4[k].y ^ rw1;
                            const int mm = __builtin_popcount(((d0 | (d0 >> 1)) | rec4[k].z) & 0x55555555u) + __builtin_popcount(((d1 | (d1 >> 1)) | rec4[k].w) & 0x55555555u);
                            cm4[k] = fig_ballot(ok4[k] && mm < mcut_cur);
                        }
                        FIG_COUNT(E, 19, lane == 0 ? 1 : 0);
                        if (!(cm4[0] | cm4[1] | cm4[2] | cm4[3])) continue;
                        for (int k = 0; k < 4; k++) {
                            unsigned long long cm = cm4[k];
                            while (cm) {
                                const int bit = fig_ctz64(cm);
                                cm &= cm - 1;
                                const int os = ob0 + k * U.wsz + bit;
                                if (os == ub.o) continue;                             // the hint, already evaluated
                                FIG_COUNT(E, 20, lane == 0 ? 1 : 0);
                                // whole-read mismatch count of placement os: lane q handles word q, then a sum over the lanes
                                const int xs = os + xoff + lwq * 16;
                                const int wj = xs >> 4, sj = (xs & 15) * 2;
                                const uint32_t e0 = kc[wj], e1 = kc[wj + 1], f0 = kn[wj], f1 = kn[wj + 1];
                                const uint32_t ev = sj ? (e0 >> sj) | (e1 << (32 - sj)) : e0, fv = sj ? (f0 >> sj) | (f1 << (32 - sj)) : f0;
                                const uint32_t dd = ev ^ rwl;
                                int mml = __builtin_popcount(((dd | (dd >> 1)) | fv) & lml);
                                int mmf = 0;
#ifdef FIG_EMU
                                mmf = 0;
                                for (int q2 = 0; q2 < nwr; q2++) {
                                    const int xq = os + xoff + q2 * 16; const int wq2 = xq >> 4, sq2 = (xq & 15) * 2;
                                    const uint32_t g0 = kc[wq2], g1 = kc[wq2 + 1], h0 = kn[wq2], h1 = kn[wq2 + 1];
                                    const uint32_t gv = sq2 ? (g0 >> sq2) | (g1 << (32 - sq2)) : g0, hv = sq2 ? (h0 >> sq2) | (h1 << (32 - sq2)) : h0;
                                    const uint32_t dq = gv ^ pk[q2]; const int remq = rs.len - q2 * 16;
                                    const uint32_t lq = remq >= 16 ? 0x55555555u : (0x55555555u >> (2 * (16 - remq)));
                                    mmf += __builtin_popcount(((dq | (dq >> 1)) | hv) & lq);
                                }
                                (void)mml;
#else
                                for (int q2 = 0; q2 < nwr; q2++) mmf += __builtin_amdgcn_readlane(mml, q2);
#endif
                                if (mmf >= mcut_cur) continue;
                                FIG_COUNT(E, 21, lane == 0 ? 1 : 0);
                                ex_acc += (unsigned long long)rs.len; const double v = fig_mle_serial<LDS>(SL, C, ncolE, pk, mt, rs.len, os + xoff, fbuf, lane, U.wsz);
                                if (v > init) { FigBest y; y.v = v; y.o = os; ub = fig_best_merge(ub, y); }
                                if (v > bound) { bound = v; mcut_cur = (int)(fig_log(bound) / lfmm) + 2; mcut_for = bound; }
                            }
                        }
                    }
                    FIG_TICK(E, 17);
                }
                for (int q = 0; q < nrounds && !swept; q++) {
                    int rr = r0 + q; if (rr >= nrounds) rr -= nrounds;
                    const int oa = obase + lane + 2 * rr * stride, ob = oa + stride;
                    const bool va_ok = oa <= w.hi, vb_ok = ob <= w.hi;
                    nplace += (va_ok ? 1 : 0) + (vb_ok ? 1 : 0);
                    // out-of-window lanes read inside the table (clamped) and are ignored
                    const int xa = (va_ok ? oa : w.hi) + xoff, xb = (vb_ok ? ob : w.hi) + xoff;
                    if (use_kf && use_serial) {
                        // mismatches of the whole read at both placements of the lane, counted in two stages: after the first
                        // 32 bases nearly every wrong placement already exceeds the cut, and the round is dropped right there
                        int mma = 0, mmb_ = 0;
                        const bool have_cut = bound > 1e-290;
                        if (have_cut && bound != mcut_for) { mcut_cur = (int)(fig_log(bound) / lfmm) + 2; mcut_for = bound; }
                        const int mcut0 = have_cut ? mcut_cur : 0x7fffffff;
                        bool dropped = false;
                        {
                            const int wia = xa >> 4, sha = (xa & 15) * 2, wib = xb >> 4, shb = (xb & 15) * 2;
                            uint32_t pca = kc[wia], pna = kn[wia], pcb = kc[wib], pnb = kn[wib];
                            // the words of the first stage are fetched together (one LDS latency instead of two)
                            const uint32_t fa1 = kc[wia + 1], ga1 = kn[wia + 1], fb1 = kc[wib + 1], gb1 = kn[wib + 1];
                            const uint32_t fa2 = kc[wia + 2], ga2 = kn[wia + 2], fb2 = kc[wib + 2], gb2 = kn[wib + 2];
                            const int nwr = (rs.len + 15) >> 4;
                            for (int wq = 0; wq < nwr; wq++) {
                                const uint32_t rw = pk[wq];
                                const int rem = rs.len - wq * 16;
                                const uint32_t lm = rem >= 16 ? 0x55555555u : (0x55555555u >> (2 * (16 - rem)));
                                uint32_t nca, nna, ncb, nnb;
                                if (wq == 0) { nca = fa1; nna = ga1; ncb = fb1; nnb = gb1; }
                                else if (wq == 1) { nca = fa2; nna = ga2; ncb = fb2; nnb = gb2; }
                                else { nca = kc[wia + wq + 1]; nna = kn[wia + wq + 1]; ncb = kc[wib + wq + 1]; nnb = kn[wib + wq + 1]; }
                                const uint32_t cva = sha ? (pca >> sha) | (nca << (32 - sha)) : pca, nva = sha ? (pna >> sha) | (nna << (32 - sha)) : pna;
                                const uint32_t cvb = shb ? (pcb >> shb) | (ncb << (32 - shb)) : pcb, nvb = shb ? (pnb >> shb) | (nnb << (32 - shb)) : pnb;
                                const uint32_t da = cva ^ rw, db = cvb ^ rw;
                                mma += __builtin_popcount(((da | (da >> 1)) | nva) & lm);
                                mmb_ += __builtin_popcount(((db | (db >> 1)) | nvb) & lm);
                                pca = nca; pna = nna; pcb = ncb; pnb = nnb;
                                if (wq == 1 && have_cut && !fig_wave_any((va_ok && mma < mcut0) || (vb_ok && mmb_ < mcut0))) { dropped = true; break; }
                            }
                        }
                        if (dropped) { FIG_TICK(E, 17); continue; }
                        if (!(bound > init)) {
                            // no maximum yet: evaluate the placement with the fewest mismatches of this round first
                            int best_mm = va_ok ? mma : 0x7fffffff, best_o = va_ok ? oa : FIG_NOPOS;
                            if (vb_ok && mmb_ < best_mm) { best_mm = mmb_; best_o = ob; }
                            FigBest bm; bm.v = -(double)best_mm; bm.o = best_o;          // larger v = fewer mismatches; ties -> smaller o
                            bm = fig_wave_best(E, bm);
                            const int o1 = fig_u(bm.o);
                            if (o1 != FIG_NOPOS) {
                                ex_acc += (unsigned long long)rs.len; const double v = fig_mle_serial<LDS>(SL, C, ncolE, pk, mt, rs.len, o1 + xoff, fbuf, lane, U.wsz);
                                if (v > init) { FigBest y; y.v = v; y.o = o1; ub = fig_best_merge(ub, y); }
                                if (v > bound) bound = v;
                            }
                        }
                        // (the bound must sit well above the denormal range: down there FP products stop shrinking -- one ulp
                        // times 0.99 rounds back to one ulp -- and fmm^m no longer bounds them; such rounds take the chain)
                        if (bound > 1e-290) {
                            // mismatch counts m with fmm^m < bound cannot reach the maximum: m >= mcut (two steps of slack
                            // cover the rounding of the logarithms)
                            if (bound != mcut_for) { mcut_cur = (int)(fig_log(bound) / lfmm) + 2; mcut_for = bound; }
                            const int mcut = mcut_cur;
                            unsigned long long sa = fig_ballot(va_ok && mma < mcut), sb = fig_ballot(vb_ok && mmb_ < mcut);
                            if (fig_popc64(sa) + fig_popc64(sb) <= 12) {
                                const int base_a = obase + 2 * rr * stride;
                                for (int half = 0; half < 2; half++) {
                                    unsigned long long m = half ? sb : sa;
                                    while (m) {
                                        int bit = 0; { unsigned long long t = m; while (!(t & 1)) { t >>= 1; bit++; } }
                                        m &= m - 1;
                                        const int os = base_a + bit + (half ? stride : 0);
                                        if (os == ub.o) continue;                   // already evaluated
                                        ex_acc += (unsigned long long)rs.len; const double v = fig_mle_serial<LDS>(SL, C, ncolE, pk, mt, rs.len, os + xoff, fbuf, lane, U.wsz);
                                        if (v > init) { FigBest y; y.v = v; y.o = os; ub = fig_best_merge(ub, y); }
                                        if (v > bound) bound = v;
                                    }
                                }
                                FIG_TICK(E, 17);
                                continue;
                            }
                        }
                    }
                    double qa = 1, qb = 1;
                    unsigned long long ma = 0, mb = 0;
                    int nb_done = 0;
#ifdef FIG_EMU
                    const int rc = fig_hot_chain_m2p<LDS, 0>(C, ncolE, pk, mt, rs.len, xa, xb, va_ok, vb_ok, mono ? bound : -FIG_DBL_MAX * 2.0, use_serial && mono, qa, qb, ma, mb, nb_done);
#else
                    const int rc = fig_hot_chain_m2p<LDS, 64>(C, ncolE, pk, mt, rs.len, oa + xoff, 0, va_ok, vb_ok, mono ? bound : -FIG_DBL_MAX * 2.0, use_serial && mono, qa, qb, ma, mb, nb_done);
#endif
                    { int steps_ = nb_done * 8; if (steps_ > rs.len) steps_ = rs.len;
                      ex_acc += (unsigned long long)steps_ * (unsigned long long)(fig_popc64(fig_ballot(va_ok)) + fig_popc64(fig_ballot(vb_ok))); }
                    FIG_TICK(E, 16);
                    if (rc == 2) {
                        if (va_ok && (qa > best.v || (qa == best.v && best.o != FIG_NOPOS && oa < best.o))) { best.v = qa; best.o = oa; }
                        if (vb_ok && (qb > best.v || (qb == best.v && best.o != FIG_NOPOS && ob < best.o))) { best.v = qb; best.o = ob; }
                        double m = fig_wave_max_dpp(best.o == FIG_NOPOS ? init : best.v);
                        if (m > bound) bound = m;
                    } else if (rc == 1) {
                        const int base_a = obase + 2 * rr * stride;
                        for (int half = 0; half < 2; half++) {
                            unsigned long long m = half ? mb : ma;
                            while (m) {
                                int bit = 0; { unsigned long long t = m; while (!(t & 1)) { t >>= 1; bit++; } }
                                m &= m - 1;
                                const int os = base_a + bit + (half ? stride : 0);
                                if (os == ub.o) continue;                       // already evaluated (the hint)
                                ex_acc += (unsigned long long)rs.len; const double v = fig_mle_serial<LDS>(SL, C, ncolE, pk, mt, rs.len, os + xoff, fbuf, lane, U.wsz);
                                if (v > init) { FigBest y; y.v = v; y.o = os; ub = fig_best_merge(ub, y); }
                                if (v > bound) bound = v;
                            }
                        }
                    }
                    FIG_TICK(E, 17);
                }
                // fold the serial results into lane 0's candidate for the wave reduction below
                if (lane == 0) best = fig_best_merge(best, ub);
                o = w.hi + 1;
            }
            for (; fig_wave_any(o <= w.hi); o += stride) {
                const bool vo = o <= w.hi;
                ex_acc += (unsigned long long)rs.len * (unsigned long long)fig_popc64(fig_ballot(vo));
                if (!vo) continue;
                double q = fig_hot_chain_m<LDS>(C, ncolE, pk, nw2, mt, rs.len, o + xoff);
                if (q > best.v) { best.v = q; best.o = o; }
                nplace++;
            }
            fl_acc += nplace * (unsigned long long)rs.len;
            best = fig_wave_best(E, best);
            // ---- the wave finishes its read (no workgroup barrier anywhere in this loop): accept test + integer pile-up
            FigBest b;
            { long long bits; memcpy(&bits, &best.v, 8); bits = fig_u64(bits); memcpy(&b.v, &bits, 8); b.o = fig_u(best.o); }
            FIG_TICK(E, 13);
            int of = b.o == FIG_NOPOS ? -left : b.o;
            if (lane == 0) E.scr.hint[r] = b.o;
            double mp = b.o == FIG_NOPOS ? init : b.v;
            double temp_log_val = -fig_log10(mp);
            bool acc = temp_log_val < U.cutoff;
            if (mode == 1) acc = acc && E.scr.saved[r] == 1;
            if (acc) {
                fig_cu32p pk = (fig_cu32p)(U.packed + rs.woff);
                int nw2 = (rs.len + 15) >> 4;
                for (int j = lane; j < rs.len; j += U.wsz) {
                    int x = of + j;
                    if (x >= 0 && x < gl && x < ncl) fig_atomic_add_i32(&nl[fig_sbase(pk, nw2, j) * nst + x], 1);
                }
            }
            if (lane == 0) {
                E.scr.accf[r] = acc ? 1 : 0;
                if (mode == 0) {
                    if (acc) {
                        E.scr.maxlv[r] = -temp_log_val;
                        fig_atomic_add_i32(&S.valid_count, 1);
                        E.scr.mark[r] = 1;
                        E.scr.frp[r * 2] = of; E.scr.frp[r * 2 + 1] = rs.len;
                        if (G == G0) { E.scr.org[r * 2] = of; E.scr.org[r * 2 + 1] = rs.len; }
                        if (G0 <= 30) {
                            int val = of + rs.len - G;
                            if (of < 0 && val > 0) { if (-of > 3 && val > 3) fig_atomic_or_i32(&S.ucoverf, 1); }
                            if (of < 0 && of + rs.len > 0) { if (-of > 3) fig_atomic_or_i32(&S.umaxleftf, 1); }
                            if (of > 0 && of < G && val > 0) { if (val > 3) fig_atomic_or_i32(&S.umaxrightf, 1); }
                        }
                    }
                } else {
                    fig_atomic_add_i32(&S.fin_i[6], 1);                                   // totalCount
                    if (acc) {
                        if (E.B->draw_pos) { E.B->draw_pos[ub + r] = of; E.B->draw_isz[ub + r] = w.tis0 + w.dir * of; }
                        E.scr.fin[r * 2] = of; E.scr.fin[r * 2 + 1] = rs.len;
                        if (of == 0) fig_atomic_or_i32(&S.fin_i[4], 1);                    // left_start_zero
                        if (of + rs.len == G) fig_atomic_or_i32(&S.fin_i[5], 1);           // right_fin_glen
                        if (of < 0 && of + rs.len > 0) { fig_atomic_or_i32(&S.fin_i[0], 1); fig_atomic_max_i32(&S.fin_i[2], -of); }
                        int val = of + rs.len - G;
                        if (of < G && val > 0) { fig_atomic_or_i32(&S.fin_i[1], 1); fig_atomic_max_i32(&S.fin_i[3], val); }
                    } else fig_atomic_add_i32(&S.fin_i[7], 1);                            // discardedCount
                }
            }
        }
        FIG_TICK(E, 3);
    }
    // ---- (4) accept test + pile-up of the reads the lane-per-read path finished (:3848-3914 / :5128-5192), one read per lane
    {
        FIG_T0(E);
        for (int r0 = 0; r0 < nU; r0 += U.nt) {
            const int r = r0 + E.tid;
            const bool mine = r < nU && mdone[r];
            if (!fig_wave_any(mine)) continue;
            bool acc = false; int of = 0, len = 0; double tlv = 0;
            fig_gcu32p pkl = (fig_gcu32p)U.packed;          // global view: a generic pointer's FLAT loads would hold up the LDS waits below
            FigWin w; w.lo = 0; w.hi = -1; w.tis0 = 0; w.dir = 1;
            if (mine) {
                len = U.u_len[ub + r];
                of = E.scr.hint[r];                                  // never FIG_NOPOS on this path
                tlv = -fig_log10(E.scr.hval[r]);
                acc = tlv < U.cutoff;
                if (mode == 1) acc = acc && E.scr.saved[r] == 1;
                pkl = (fig_gcu32p)U.packed + U.u_woff[ub + r];
                w = fig_window_u(U, U.u_pos[ub + r], len, gapoffset);
                E.scr.accf[r] = acc ? 1 : 0;
            }
            {   // integer pile-up (order-free)
                const int nw2 = (len + 15) >> 4;
                for (int wi = 0; fig_wave_any(acc && wi < nw2); wi++) {
                    if (!(acc && wi < nw2)) continue;
                    const uint32_t wd = pkl[wi];
                    int nb = len - wi * 16; if (nb > 16) nb = 16;
                    for (int jj = 0; jj < nb; jj++) {
                        const int x = of + wi * 16 + jj;
                        if (x >= 0 && x < gl && x < ncl) fig_atomic_add_i32(&nl[(int)((wd >> (2 * jj)) & 3) * nst + x], 1);
                    }
                }
            }
            if (mode == 0) {
                const int nacc = fig_popc64(fig_ballot(acc));
                if (lane == 0 && nacc) fig_atomic_add_i32(&S.valid_count, nacc);
                if (acc) {
                    E.scr.maxlv[r] = -tlv;
                    E.scr.mark[r] = 1;
                    E.scr.frp[r * 2] = of; E.scr.frp[r * 2 + 1] = len;
                    if (G == G0) { E.scr.org[r * 2] = of; E.scr.org[r * 2 + 1] = len; }
                    if (G0 <= 30) {
                        const int val = of + len - G;
                        if (of < 0 && val > 0) { if (-of > 3 && val > 3) fig_atomic_or_i32(&S.ucoverf, 1); }
                        if (of < 0 && of + len > 0) { if (-of > 3) fig_atomic_or_i32(&S.umaxleftf, 1); }
                        if (of > 0 && of < G && val > 0) { if (val > 3) fig_atomic_or_i32(&S.umaxrightf, 1); }
                    }
                }
            } else {
                const int nm = fig_popc64(fig_ballot(mine)), nrej = fig_popc64(fig_ballot(mine && !acc));
                if (lane == 0) { fig_atomic_add_i32(&S.fin_i[6], nm); if (nrej) fig_atomic_add_i32(&S.fin_i[7], nrej); }
                if (acc) {
                    if (E.B->draw_pos) { E.B->draw_pos[ub + r] = of; E.B->draw_isz[ub + r] = w.tis0 + w.dir * of; }
                    E.scr.fin[r * 2] = of; E.scr.fin[r * 2 + 1] = len;
                    if (of == 0) fig_atomic_or_i32(&S.fin_i[4], 1);
                    if (of + len == G) fig_atomic_or_i32(&S.fin_i[5], 1);
                    if (of < 0 && of + len > 0) { fig_atomic_or_i32(&S.fin_i[0], 1); fig_atomic_max_i32(&S.fin_i[2], -of); }
                    const int val = of + len - G;
                    if (of < G && val > 0) { fig_atomic_or_i32(&S.fin_i[1], 1); fig_atomic_max_i32(&S.fin_i[3], val); }
                }
            }
        }
        FIG_TICK(E, 26);
    }
    E.flops += fl_acc; E.mle_alg += fl_acc; E.mle_exec += ex_acc;
    FIG_SYNC();
    if (nci_lds) {
        for (int x = E.tid; x < ncl; x += E.nt) for (int j = 0; j < 5; j++) E.scr.nci[j * cg + x] = nl[j * nst + x];
        FIG_SYNC();
    }
}

// The MLE pass behind its class's memory form.  LDS-tiled class: the E-step's image and weight rows are idle now, and the
// MLE table needs 5 doubles per column where {P,Q}+Q4 needed 9, so the pass runs its LDS form over the whole column range
// whenever C[5][ncolE] + the per-column records + its buffers fit the area in front of FigState (the packer sized the
// workgroup's LDS for it when it could); otherwise it runs on the table in the scratch slab.
template <bool LDS>
FIG_D void fig_mle_dispatch(FigEng &E, int gapoffset, int mode, int gl, int wl, int wr) {
    if (!(LDS && E.tiles > 0)) { fig_hot_mle<LDS>(E, gapoffset, mode, gl, wl, wr); return; }
    if (FIG_TILED_MLE_DOUBLES(E.ncolE, E.nw) > E.lds_tw) { fig_hot_mle<false>(E, gapoffset, mode, gl, wl, wr); return; }
    FigPQ *const pq_keep = E.pq; double *const q4_keep = E.q4; double *const w_keep = E.wbuf;
    const int off_w_keep = E.off_w, nteams_keep = E.nteams, Wcap_keep = E.Wcap;
    E.pq = (FigPQ *)(fig_lds + E.off_pq); E.q4 = nullptr; E.off_w = E.off_pq + 7 * E.ncolE; E.wbuf = fig_lds + E.off_w;
    E.nteams = 1; E.Wcap = E.lds_tw - 7 * E.ncolE;              // "one weight row" = everything behind the records
    fig_hot_mle<true>(E, gapoffset, mode, gl, wl, wr);
    E.pq = pq_keep; E.q4 = q4_keep; E.wbuf = w_keep; E.off_w = off_w_keep; E.nteams = nteams_keep; E.Wcap = Wcap_keep;
}

#endif
