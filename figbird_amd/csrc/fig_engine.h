// fig_engine.h -- the per-gap EM gap-filling engine for gfx950 (MI355X).
//
// One workgroup owns one gap at a time and runs the whole of what the reference does per gap
// (GapFiller::fillGap, Figbird.cpp:6201-6570) without leaving the device: candidate-length
// loop x EM loop x placeReads (E-step pile-up + MLE pass), consensus, the unmapped/partial
// post-processing heuristics and finalize.  Workgroups pull gaps from a cost-sorted queue.
//
// Parallel structure inside a workgroup (nt = blockDim.x lanes, 64..1024):
//   * E-step (A4): lanes = placements of the current read; each lane runs the ordered
//     product chain over the read's bases; the per-placement weight goes to LDS (wbuf);
//     then lanes = gap columns, each column adding the weights that cover it in exactly the
//     (read, placement) order of the reference, so countsGap is bit-identical (no atomics).
//   * MLE pass (A5/A12): lanes = placements, workgroup arg-max ("first maximum wins").
//   * computeProbsGap / computeErrorProbsGap / computeSequence (A3/A6): lanes = gap columns.
//   * control logic and the integer/string heuristics (A7-A11): lane 0, between barriers.
// LDS holds the per-column probability tables P[4][G], Q[5][G] (72 B/column), the weight
// buffer, the staged read and the consensus codes; flank columns are one-hot and are kept
// as byte codes plus a 6x5 table.  Everything else is in a per-workgroup HBM/L2 scratch slab.
//
// The same source also compiles under FIG_EMU (tests/emu) as a ONE-LANE host emulation used
// only by the CPU unit tests to check the control logic against the oracle; it is not a
// fallback and is never linked into libfighip.so.
#ifndef FIG_ENGINE_H
#define FIG_ENGINE_H

#include "fig_types.h"

#ifdef FIG_EMU
#include <cmath>
#include <cstring>
#define FIG_D static inline
#define FIG_NOINLINE
#define FIG_HD static inline
#define FIG_SYNC() ((void)0)
FIG_D double fig_log10(double x) { return log10(x); }
FIG_D double fig_log(double x) { return log(x); }
FIG_D double fig_exp(double x) { return exp(x); }
FIG_D double fig_pow10(double x) { return pow(10, x); }
FIG_D int fig_atomic_add_i32(int *p, int v) { int o = *p; *p += v; return o; }
FIG_D void fig_atomic_max_i32(int *p, int v) { if (v > *p) *p = v; }
FIG_D int fig_atomic_fetch_add_i32(int *p, int v) { int o = *p; *p += v; return o; }
FIG_D void fig_atomic_min_i32(int *p, int v) { if (v < *p) *p = v; }
FIG_D void fig_atomic_or_i32(int *p, int v) { *p |= v; }
FIG_D void fig_atomic_add_u64(unsigned long long *p, unsigned long long v) { *p += v; }
FIG_D unsigned long long fig_shfl_down_u64(unsigned long long v, int off) { (void)v; (void)off; return 0; }   // one lane: nothing to the right
#else
#include <hip/hip_runtime.h>
#define FIG_D __device__ static
// kept out of line on purpose: a dispatcher that inlines one of its big callees uses all 256 VGPRs itself and then saves and
// restores ~100 callee-saved registers per lane on every call (fig_hot_estep_dispatch did so with the tiled variants)
#define FIG_NOINLINE __attribute__((noinline))
#define FIG_HD __host__ __device__ static
#ifdef FIG_PROF
// diagnostic build: every lane accumulates the cycles it spends in workgroup barriers (E must be in scope)
#define FIG_SYNC() do { unsigned long long _bt = __builtin_readcyclecounter(); __syncthreads(); E.wait_cycles += __builtin_readcyclecounter() - _bt; } while (0)
#else
#define FIG_SYNC() __syncthreads()
#endif
FIG_D double fig_log10(double x) { return log10(x); }
FIG_D double fig_log(double x) { return log(x); }
FIG_D double fig_exp(double x) { return exp(x); }
FIG_D double fig_pow10(double x) { return pow(10.0, x); }
FIG_D int fig_atomic_add_i32(int *p, int v) { return atomicAdd(p, v); }
FIG_D void fig_atomic_max_i32(int *p, int v) { atomicMax(p, v); }
FIG_D int fig_atomic_fetch_add_i32(int *p, int v) { return atomicAdd(p, v); }
FIG_D void fig_atomic_min_i32(int *p, int v) { atomicMin(p, v); }
FIG_D void fig_atomic_or_i32(int *p, int v) { atomicOr(p, v); }
FIG_D void fig_atomic_add_u64(unsigned long long *p, unsigned long long v) { atomicAdd(p, v); }
FIG_D unsigned long long fig_shfl_down_u64(unsigned long long v, int off) { return __shfl_down(v, off, 64); }
#endif

#define FIG_NOPOS 0x7fffffff
#define FIG_PLB_TEAMS 8
#define FIG_PLB_BYTES (FIG_PLB_TEAMS * 64 * 4)

// Optional phase timers (diagnostic build only: -DFIG_PROF).  Every lane accumulates s_memtime deltas in its own
// FigEng::prof[] (no atomics inside the loops); lane 0 of each wave adds them to B.counters[8+slot] when the kernel
// ends, so the printed figures are sums over WAVES.  Never enabled in the shipped library.
#if defined(FIG_PROF) && !defined(FIG_EMU)
#define FIG_T0(E) unsigned long long _fig_t = __builtin_readcyclecounter()
#define FIG_TICK(E, slot) do { unsigned long long _n = __builtin_readcyclecounter(); (E).prof[slot] += _n - _fig_t; _fig_t = _n; } while (0)
#define FIG_COUNT(E, slot, n) do { (E).prof[slot] += (n); } while (0)
#else
#define FIG_COUNT(E, slot, n) do { } while (0)
#define FIG_T0(E) do { } while (0)
#define FIG_TICK(E, slot) do { } while (0)
#endif

// LDS base of the workgroup.  On the device it is the dynamic shared segment, so pointers formed
// from it keep address space 3 (ds_read/ds_write); under FIG_EMU it is a host buffer.
#ifdef FIG_EMU
static double *fig_lds = nullptr;
#define FIG_RFL(x) (x)
typedef const uint32_t *fig_cu32p;
typedef const double *fig_cdp;
typedef double *fig_gdp;
typedef const double *fig_gcdp;
typedef const uint32_t *fig_gcu32p;
typedef const unsigned char *fig_gcu8p;
typedef const unsigned char *fig_lcu8p;
#else
extern __shared__ __attribute__((aligned(16))) double fig_lds[];
#define FIG_RFL(x) __builtin_amdgcn_readfirstlane(x)
// constant address space: uniform loads through these become scalar (s_load) instructions
typedef const uint32_t __attribute__((address_space(4))) *fig_cu32p;
typedef const double __attribute__((address_space(4))) *fig_cdp;
// Global (address space 1) views of slab and batch arrays.  fig_uptr() hands out generic pointers, and a generic access is a FLAT
// instruction: it counts on lgkmcnt as well as vmcnt and returns out of order with LDS operations, so with one in flight every LDS
// wait becomes lgkmcnt(0) and also waits for the global round trip.  global_load / global_store count on vmcnt alone.
typedef double __attribute__((address_space(1))) *fig_gdp;
typedef const double __attribute__((address_space(1))) *fig_gcdp;
typedef const uint32_t __attribute__((address_space(1))) *fig_gcu32p;
typedef const unsigned char __attribute__((address_space(1))) *fig_gcu8p;     // batch byte arrays (flank codes)
typedef const unsigned char __attribute__((address_space(3))) *fig_lcu8p;     // byte strings behind FigState in LDS (E.gs)
#endif

// 16-byte aligned so that the hot loop fetches a {P,Q} pair with ONE ds_read_b128 (256 B/clk/CU);
// the ds_read2_b64 the compiler emits for an 8-byte-aligned pair is 2-way bank-conflicted at this stride.
struct alignas(16) FigPQ { double p, q; };       // probsGap[x][b], errorProbsGap[x][b] of one (column, base)

// Loop-carried variables of fillGap's candidate-length loop (Figbird.cpp:6237-6296), kept in memory so the
// loop can be cut into pieces: begin (setup + checkGapReads), per-candidate evaluation, per-candidate
// bookkeeping ("step"), end (fallbacks + finalize).  Sequential mode runs the pieces back to back in one
// workgroup; candidate-parallel mode evaluates several candidates of a gap in different workgroups and
// replays the bookkeeping in order.
struct FigLoop {
    int inr, range, gapMin, j, gapEstimate, maxGapEstimate, finalize_flag;
    int fill_or_not, same_count, same_thresh, stuckCount, side_flag, less_read_flag;
    int prev_best, curr_best, prev_u, curr_u, sec_same, sec_same2;
    int done, need_orig;
    // result of the last evaluated candidate (what the bookkeeping step consumes)
    int ev_iters, ev_side_break, ev_fill;
    double maxLikelihood, secondMaxLikelihood, prevlikelihood, likelihood;
};

// ---------------------------------------------------------------------------------------
// Workgroup-shared scalar state (LDS).  Names follow GapFiller's members (Figbird.cpp:1563-1635).
struct FigState {
    int G, ncols;                    // this->gapLength ; columns spanned at the last initialize_start_end
    int left, right;                 // left_maxDistance, right_maxDistance (file-level globals in the reference)
    int side_limit; long long end_pos_max;
    int valid_count, invalid_count, partial_read_count;
    double region_perct, region_perct_max;
    int partial_read_len, rep_flag, one_side_repeat_flag, large_gap_flag, comp_count;
    int left_max, right_min, discont_or_not;
    int psr_temp[2], psr_final[2];
    int umaxleftf, umaxrightf, ucoverf;
    int num_itr, overlap_threshold, gaptofill;
    int gl_len, gr_len, pl_len, pr_len;
    int cons_len, best_len, cur_len, prev_len, orig_len;
    unsigned char gap_left[32], gap_right[32];
    unsigned char partial_left[100], partial_right[100];
    double FP[6][5], FQ[6][5];       // flank-column tables by flank code (0..3 one-hot, 4 = N, 5 = outside window)
    // reductions / broadcast
    double red_v[16]; int red_i[16];
    double bc_d; int bc_i, bc_j;
    int ibuf[8];
    double lik;                      // return value of placeReads / run
    int ctl[8];
    // read teams of the unmapped hot path: one read per team, T waves per team
    int tm_lo[16], tm_hi[16], tm_len[16], tm_tis0[16], tm_dir[16];
    int tm_aux[16]; long long tm_woff[16];   // LDS-tiled class: the chunk's read scalars, staged once per chunk
    double wv_v[32]; int wv_o[32];   // per-wave partial arg-max ([team][wave of the team]; LDS-tiled class: [read of the chunk][wave])
    // shared-factor E-step (fig_engine_shared.h): insert-size windows of the chunk's 32 reads; reads that take the generic chain
    short sh_lo[FIG_SH_SC * FIG_SH_C], sh_hi[FIG_SH_SC * FIG_SH_C];      // (the shared form is taken for <= 1024 placements only)
    int sh_tis0[FIG_SH_SC * FIG_SH_C];
    unsigned sh_woff[FIG_SH_SC * FIG_SH_C];                              // packed-read word offset relative to the gap's first read
    unsigned char sh_len[FIG_SH_SC * FIG_SH_C], sh_aux[FIG_SH_SC * FIG_SH_C];   // aux: bit 0 reverse, bit 1 N base, bit 2 window direction +1
    unsigned sh_irr[16], sh_pad;        // [chunk of the super-chunk] (two per staging wave)
    int mle_next, pad_mn;            // next read of the MLE pass (waves take reads dynamically)
    int fin_i[8];                    // finalize statistics (flags / max / counts), filled with atomics
    FigLoop L;
    // useful-work counters of this gap (speculative candidates that are discarded never reach them)
    unsigned long long flops_useful; int n_place, pad_np;
    int dbg_j, pad_dj;               // candidate index whose numeric planes are being exported (-1: none)
};

struct FigTrip { int v[3]; };



// Per-workgroup scratch slab (HBM, L2-resident while the gap is being worked on).
struct FigScr {
    double *cnt;       // countsGap      [5][capG]   (gap columns only)
    double *ncnt;      // new_counts_gap [5][capG]
    double *tmp;       // qual_gap (partial) / count_pos (seed re-weighting) [5][capG]
    int *pc;           // partial_count_array [4][capG]
    int *nci;          // integer pile-up of accepted reads [5][capG] (new_counts_gap / finalize's countsGap increments)
    unsigned char *accf; // [R] accepted flag of the current MLE pass
    unsigned char *mdone; // [R] the lane-per-read MLE path finished this read's search (fig_hot_mle)
    int *cov;          // gap_coverage [capG]
    int *region;       // [capG+8]
    unsigned char *cons, *best, *cur, *prev, *orig;   // strings [capG+1]
    unsigned char *colchar;                           // [capG] per-column majority code (update_partial_prob)
    // per unmapped read
    unsigned char *mark, *saved;
    double *maxlv;
    int *frp;          // final_readpos [R][2] (pos, len) ; id is the index
    int *org;          // unmapped_read_pos_arr_org [R][2]
    int *fin;          // finalize's unmapped_read_pos_arr [R][2]
    FigTrip *sortbuf;  // [R]
    int *hint;         // [R] best placement of the previous MLE pass over this read (pruning hint only; never affects results)
    int *hint_e;       // [R] arg-max placement of the last E-step over this read (second pruning hint)
    double *hval;      // [R] MLE product at the chosen hint, evaluated lane-per-read before the per-read loop (-1: not evaluated)
    // per partial read
    int *repeatflag;   // [P][3]
    int *ppos_org;     // partial_read_pos_arr_org [P][3]
    int *pflag;        // [P][2]
    int *prf;          // partial_read_flag [P][3]
    int *lcross, *rcross; unsigned char *smflag;      // [P]
    // per candidate
    int *used_read_arr; int *lrmd;                    // [C], [C][2]
    FigPQ *pqg; double *q4g;   // extended probability table when it does not fit in LDS
    double *wg;        // global weight buffer when it does not fit in LDS
};

FIG_HD long long fig_align8(long long x) { return (x + 15) & ~15LL; }   // (16: FigPQ rows want ds_read_b128 / dwordx4 alignment)

// Carve the slab.  Returns the total size when base == nullptr.
// capG = scratch columns (max alloc_arg), capE = extended table columns, capW = weight-buffer doubles
FIG_HD long long fig_scratch_layout(unsigned char *base, int capG, int capR, int capP, int capC, int capW, int capE, FigScr *s) {
    long long o = 0;
#define FIG_CARVE(field, type, count) do { if (s) s->field = (type *)(base + o); o = fig_align8(o + (long long)sizeof(type) * (count)); } while (0)
    FIG_CARVE(cnt, double, 5LL * capG);
    FIG_CARVE(ncnt, double, 5LL * capG);
    FIG_CARVE(tmp, double, 5LL * capG);
    FIG_CARVE(pqg, FigPQ, 4LL * capE);
    FIG_CARVE(q4g, double, capE);
    FIG_CARVE(wg, double, capW);
    FIG_CARVE(maxlv, double, capR > capP ? capR : capP);
    FIG_CARVE(hval, double, capR);
    FIG_CARVE(pc, int, 4LL * capG);
    FIG_CARVE(nci, int, 5LL * capG);
    FIG_CARVE(cov, int, capG);
    FIG_CARVE(region, int, 2LL * capG + 16);
    FIG_CARVE(frp, int, 2LL * capR);
    FIG_CARVE(org, int, 2LL * capR);
    FIG_CARVE(fin, int, 2LL * capR);
    FIG_CARVE(sortbuf, FigTrip, capR);
    FIG_CARVE(hint, int, capR);
    FIG_CARVE(hint_e, int, capR);
    FIG_CARVE(repeatflag, int, 3LL * capP);
    FIG_CARVE(ppos_org, int, 3LL * capP);
    FIG_CARVE(pflag, int, 2LL * capP);
    FIG_CARVE(prf, int, 3LL * capP);
    FIG_CARVE(lcross, int, capP);
    FIG_CARVE(rcross, int, capP);
    FIG_CARVE(used_read_arr, int, capC);
    FIG_CARVE(lrmd, int, 2LL * capC);
    FIG_CARVE(cons, unsigned char, capG + 8);
    FIG_CARVE(best, unsigned char, capG + 8);
    FIG_CARVE(cur, unsigned char, capG + 8);
    FIG_CARVE(prev, unsigned char, capG + 8);
    FIG_CARVE(orig, unsigned char, capG + 8);
    FIG_CARVE(colchar, unsigned char, capG + 8);
    FIG_CARVE(mark, unsigned char, capR + 8);
    FIG_CARVE(saved, unsigned char, capR + 8);
    FIG_CARVE(accf, unsigned char, capR + 8);
    FIG_CARVE(mdone, unsigned char, capR + 8);
    FIG_CARVE(smflag, unsigned char, capP + 8);
#undef FIG_CARVE
    return o;
}

// Everything a lane needs; lives in registers / private memory.
struct FigEng {
    int tid, nt;
    const FigDevModel *M;
    const FigDevBatch *B;
    const FigDevGap *g;
    FigState *S;
    FigScr scr;
    // Extended probability table: columns x in [-(L-1), Gmax+L-1) are stored at x+xoff.  pq[b*ncolE + xe]
    // holds {probsGap, errorProbsGap} of base b<4; q4[xe] = errorProbsGap[..][4].  Flank columns are
    // materialised from the one-hot tables so the E-step inner loop is branch-free.
    FigPQ *pq; double *q4;           // generic pointers (LDS or scratch)
    int ncolE, xoff;
    int off_pq, off_q4, off_w;       // offsets (in doubles) from fig_lds when the table / weights live in LDS
    int off_plb;                     // offset (in doubles) of plb from fig_lds (always LDS)
    int pq_lds, w_lds;
    int lds_tw;                      // doubles of LDS in front of FigState (table/image + weight rows)
    int tiles, tile_step, tile_cols; // > 0: the table lives in HBM/L2 (pq, q4) and the E-step streams it through an LDS tile at off_pq / off_q4
    double *wbuf;                    // [nteams][Wcap]
    int Wcap, nteams;
    int sh_on;                       // the shared-factor E-step may be used (FIG_ESTEP=pair switches it off)
    int lane, wave, nw, wsz;         // lane in wave, wave in workgroup, waves per workgroup, lanes per wave
    const double *kt_fwd, *kt_rev;   // {1-e[k], e[k]} pairs, forward and reversed (index (L-len)+j), for scalar loads
    const double *mt_fwd, *mt_rev;   // {1-e-ins-del, e[k]} pairs for the MLE pass
    unsigned char *rb;               // staged read codes [FIG_MAX_READLEN + 8]
    uint32_t *plb;                   // per-team staged position lists of the chunk's reads [FIG_PLB_TEAMS][64] (LDS)
    unsigned char *gs;               // consensus codes of the gap columns for the MLE pass [capG]
    int capG;
    unsigned long long flops;        // per-lane algorithmic flop count
    unsigned long long mle_alg;      // per-lane share of `flops` credited to the MLE passes (1 per placement and base)
    unsigned long long mle_exec;     // FP64 multiplies the MLE passes actually executed after pruning (wave total, kept in lane 0)
    unsigned long long wait_cycles;  // FIG_PROF only: cycles spent in workgroup barriers
    unsigned long long prof[40];     // FIG_PROF only: phase timers
};

// ---------------------------------------------------------------------------------------
// packed read access
FIG_D int fig_read_code(const uint32_t *packed, long long woff, int len, int j) {
    int nw2 = (len + 15) >> 4;
    uint32_t w = packed[woff + (j >> 4)];
    uint32_t m = packed[woff + nw2 + (j >> 5)];
    int c = (w >> ((j & 15) * 2)) & 3;
    return ((m >> (j & 31)) & 1) ? 4 : c;
}

FIG_D int fig_flank_l(const FigEng &E, int k) {      // base at gapStart-k, k >= 1
    if (k > FIG_FLANK) return 4;
    return ((fig_gcu8p)E.B->flank)[E.g->flankOff + (k - 1)];
}
FIG_D int fig_flank_r(const FigEng &E, int k) {      // base at gapStart+G0+k, k >= 0
    if (k >= FIG_FLANK) return 4;
    return ((fig_gcu8p)E.B->flank)[E.g->flankOff + FIG_FLANK + k];
}

// Stage read (unmapped u / partial p) into LDS as byte codes.  Caller syncs afterwards.
FIG_D void fig_stage_read(FigEng &E, const FigDevReads &R, long long idx) {
    int len = R.len[idx];
    long long woff = R.woff[idx];
    for (int j = E.tid; j < len; j += E.nt) E.rb[j] = (unsigned char)fig_read_code(E.B->packed, woff, len, j);
}

// Private-memory copy of a read (serial string heuristics).
FIG_D int fig_load_read(const FigEng &E, const FigDevReads &R, long long idx, unsigned char *dst) {
    int len = R.len[idx];
    long long woff = R.woff[idx];
    for (int j = 0; j < len; j++) dst[j] = (unsigned char)fig_read_code(E.B->packed, woff, len, j);
    return len;
}

// ---------------------------------------------------------------------------------------
// workgroup reductions
struct FigBest { double v; int o; };

// countsGap additions of one placement: read bases that land inside the gap columns [0, G) (Figbird.cpp:3603-3611)
FIG_D int fig_ovl(int o, int len, int G) { int a = o > 0 ? o : 0, b = o + len < G ? o + len : G; return b > a ? b - a : 0; }

// "first maximum wins": larger v, ties -> smaller o.  Entries with o == FIG_NOPOS are empty.
FIG_D FigBest fig_best_merge(FigBest a, FigBest b) {
    if (b.o == FIG_NOPOS) return a;
    if (a.o == FIG_NOPOS) return b;
    if (b.v > a.v || (b.v == a.v && b.o < a.o)) return b;
    return a;
}

FIG_D FigBest fig_block_best(FigEng &E, FigBest x) {
#ifdef FIG_EMU
    return x;
#else
    for (int off = 32; off > 0; off >>= 1) {
        FigBest y;
        y.v = __shfl_down(x.v, off, 64);
        y.o = __shfl_down(x.o, off, 64);
        x = fig_best_merge(x, y);
    }
    int wave = E.tid >> 6, nw = (E.nt + 63) >> 6;
    if (nw == 1) {
        x.v = __shfl(x.v, 0, 64);
        x.o = __shfl(x.o, 0, 64);
        return x;
    }
    FIG_SYNC();                       // previous users of red_* are done
    if ((E.tid & 63) == 0) { E.S->red_v[wave] = x.v; E.S->red_i[wave] = x.o; }
    FIG_SYNC();
    FigBest r; r.v = E.S->red_v[0]; r.o = E.S->red_i[0];
    for (int w = 1; w < nw; w++) { FigBest y; y.v = E.S->red_v[w]; y.o = E.S->red_i[w]; r = fig_best_merge(r, y); }
    return r;
#endif
}

// ---------------------------------------------------------------------------------------
// initialize_start_end, Figbird.cpp:2269-2296 (lane 0)
FIG_D void fig_ise(FigEng &E) {
    FigState &S = *E.S;
    long long gs = E.g->gapStart, cl = E.g->contigLen;
    if (gs - S.left < 0) { S.left = (int)gs; if (S.side_limit > S.left) S.side_limit = S.left; }
    if (gs + S.G + S.right > cl) { S.right = (int)(cl - (gs + S.G)); if (S.side_limit > S.right) S.side_limit = S.right; }
    long long endPos = gs + S.G + S.right;
    if (endPos > S.end_pos_max) S.end_pos_max = endPos;
    S.ncols = S.G;
}

// Flank-column probability rows (computeProbsGap + computeErrorProbsGap on a one-hot /
// N column, Figbird.cpp:2094-2109, 2122-2136).  Lane 0, once per gap.
FIG_D void fig_flank_tables(FigEng &E) {
    FigState &S = *E.S;
    for (int c = 0; c < 6; c++) {
        double cnt[5] = {0, 0, 0, 0, 0};
        if (c < 5) cnt[c] = 1;
        double total = 0;
        for (int j = 0; j < 5; j++) total = total + cnt[j];
        double Ncount = cnt[4];
        double pr[5];
        for (int j = 0; j < 4; j++) pr[j] = (c < 5) ? ((cnt[j] + (Ncount / 4)) / total) : 0.0;
        pr[4] = 0;
        for (int j = 0; j < 5; j++) {
            double sum = 0;
            for (int k = 0; k < 4; k++) { if (j == k) continue; sum += pr[k] * E.M->T[k * 5 + j]; }
            S.FQ[c][j] = (c < 5) ? sum : 0.0;
            S.FP[c][j] = pr[j];
        }
    }
}

// Code of window column x (gap-relative) for the E-step: -1 = left of the window (skipped,
// `index<0`, Figbird.cpp:3578), 0..4 flank base, 5 = right of the window (all-zero row), 6 = gap column.
FIG_D int fig_col_kind(const FigEng &E, int x, int G, int left, int right) {
    if (x < 0) { if (x < -left) return -1; return fig_flank_l(E, -x); }
    if (x < G) return 6;
    if (x < G + right) return fig_flank_r(E, x - G);
    return 5;
}
// charCode of gapString[index] for the MLE pass (Figbird.cpp:3765); outside the string -> 4.
FIG_D int fig_from_code(const FigEng &E, int x, int G, int left, int right) {
    if (x < 0) { if (x < -left) return 4; return fig_flank_l(E, -x); }
    if (x < G) return ((fig_lcu8p)E.gs)[x];
    if (x < G + right) return fig_flank_r(E, x - G);
    return 4;
}

// fig_from_code on hoisted, typed views: fl = the gap's flank codes (global), gs = the gap string (LDS).  The MLE pass packs the
// consensus of every extended column with it, 16 columns per word: through fig_from_code each of those lookups reloads the
// descriptor fields behind E.B / E.g with FLAT instructions.
FIG_D int fig_from_code_u(fig_gcu8p fl, fig_lcu8p gs, int x, int G, int left, int right) {
    if (x < 0) { if (x < -left) return 4; const int k = -x; return k > FIG_FLANK ? 4 : fl[k - 1]; }
    if (x < G) return gs[x];
    if (x < G + right) { const int k = x - G; return k >= FIG_FLANK ? 4 : fl[FIG_FLANK + k]; }
    return 4;
}

// ---------------------------------------------------------------------------------------
// Accessors of the extended table (generic pointers; the hot loops use the typed LDS form).
#define FIG_PQ(E, b, x) ((E).pq[(long long)(b) * (E).ncolE + (x) + (E).xoff])
#define FIG_Q4(E, x) ((E).q4[(x) + (E).xoff])
// The same accessors on typed views, selected by E.pq_lds (wave-uniform): ds_* / global_* instead of FLAT instructions in the
// per-candidate and per-call table code (initialize, computeProbsGap, computeErrorProbsGap, update_partial_prob).
#ifdef FIG_EMU
FIG_D void fig_pq_set(const FigEng &E, int b, int x, FigPQ v) { FIG_PQ(E, b, x) = v; }
FIG_D FigPQ fig_pq_get(const FigEng &E, int b, int x) { return FIG_PQ(E, b, x); }
FIG_D void fig_pq_set_p(const FigEng &E, int b, int x, double v) { FIG_PQ(E, b, x).p = v; }
FIG_D void fig_pq_set_q(const FigEng &E, int b, int x, double v) { FIG_PQ(E, b, x).q = v; }
FIG_D double fig_pq_get_p(const FigEng &E, int b, int x) { return FIG_PQ(E, b, x).p; }
FIG_D void fig_q4_set(const FigEng &E, int x, double v) { FIG_Q4(E, x) = v; }
FIG_D double fig_q4_get(const FigEng &E, int x) { return FIG_Q4(E, x); }
#else
typedef double fig_d2v __attribute__((ext_vector_type(2)));          // one {p, q} entry as a 16-byte vector (ds_read/write_b128, global dwordx4)
typedef fig_d2v __attribute__((address_space(3))) *fig_ld2p;
typedef fig_d2v __attribute__((address_space(1))) *fig_gd2p;
typedef double __attribute__((address_space(3))) *fig_ldp;
FIG_D void fig_pq_set(const FigEng &E, int b, int x, FigPQ v) {
    const long long i = (long long)b * E.ncolE + x + E.xoff;
    fig_d2v t; t.x = v.p; t.y = v.q;
    if (E.pq_lds) ((fig_ld2p)(fig_lds + E.off_pq))[i] = t; else ((fig_gd2p)E.pq)[i] = t;
}
FIG_D FigPQ fig_pq_get(const FigEng &E, int b, int x) {
    const long long i = (long long)b * E.ncolE + x + E.xoff;
    fig_d2v t;
    if (E.pq_lds) t = ((fig_ld2p)(fig_lds + E.off_pq))[i]; else t = ((fig_gd2p)E.pq)[i];
    FigPQ v; v.p = t.x; v.q = t.y;
    return v;
}
FIG_D void fig_pq_set_p(const FigEng &E, int b, int x, double v) {
    const long long i = (long long)b * E.ncolE + x + E.xoff;
    if (E.pq_lds) ((fig_ldp)(fig_lds + E.off_pq))[2 * i] = v; else ((fig_gdp)(double *)E.pq)[2 * i] = v;
}
FIG_D void fig_pq_set_q(const FigEng &E, int b, int x, double v) {
    const long long i = (long long)b * E.ncolE + x + E.xoff;
    if (E.pq_lds) ((fig_ldp)(fig_lds + E.off_pq))[2 * i + 1] = v; else ((fig_gdp)(double *)E.pq)[2 * i + 1] = v;
}
FIG_D double fig_pq_get_p(const FigEng &E, int b, int x) {
    const long long i = (long long)b * E.ncolE + x + E.xoff;
    if (E.pq_lds) return ((fig_ldp)(fig_lds + E.off_pq))[2 * i];
    return ((fig_gdp)(double *)E.pq)[2 * i];
}
FIG_D void fig_q4_set(const FigEng &E, int x, double v) {
    if (E.pq_lds) ((fig_ldp)(fig_lds + E.off_q4))[x + E.xoff] = v; else ((fig_gdp)E.q4)[x + E.xoff] = v;
}
FIG_D double fig_q4_get(const FigEng &E, int x) {
    if (E.pq_lds) return ((fig_ldp)(fig_lds + E.off_q4))[x + E.xoff];
    return ((fig_gdp)E.q4)[x + E.xoff];
}
#endif

// Flank columns of the extended table for the current window (G, left, right):
//   x in [-left, 0)      left-flank base at gapStart+x          -> one-hot / N rows of FP,FQ
//   x in [G, G+right)    right-flank base at gapStart+G0+(x-G)
//   anything else        all-zero row (the reference never initialises those rows; placements that start
//                        left of the window skip them via `index<0`, Figbird.cpp:3578)
// All lanes; caller syncs.
FIG_D void fig_build_flank_pq(FigEng &E) {
    const FigState &S = *E.S;
    int G = S.G, left = S.left, right = S.right, L1 = E.xoff;
    int n = 2 * L1;
    for (int i = E.tid; i < n; i += E.nt) {
        int x = i < L1 ? i - L1 : G + (i - L1);
        if (x + E.xoff >= E.ncolE) continue;
        int c;
        if (x < 0) c = (-x <= left) ? fig_flank_l(E, -x) : 5;
        else c = (x - G < right) ? fig_flank_r(E, x - G) : 5;
        for (int b = 0; b < 4; b++) { FigPQ v; v.p = S.FP[c][b]; v.q = S.FQ[c][b]; fig_pq_set(E, b, x, v); }
        fig_q4_set(E, x, S.FQ[c][4]);
    }
}

// computeProbsGap(0) + computeErrorProbsGap on the gap columns (A3), Figbird.cpp:2090-2137.
// All lanes; caller syncs.
FIG_D void fig_compute_probs(FigEng &E) {
    int n = E.S->ncols, cg = E.capG;
    const double *T = E.M->T;
    for (int x = E.tid; x < n; x += E.nt) {
        double c0 = E.scr.cnt[x], c1 = E.scr.cnt[cg + x], c2 = E.scr.cnt[2 * cg + x], c3 = E.scr.cnt[3 * cg + x], c4 = E.scr.cnt[4 * cg + x];
        double total = 0;
        total = total + c0; total = total + c1; total = total + c2; total = total + c3; total = total + c4;
        double pr[4];
        if (total) {
            double nq = c4 / 4;
            pr[0] = (c0 + nq) / total; pr[1] = (c1 + nq) / total; pr[2] = (c2 + nq) / total; pr[3] = (c3 + nq) / total;
        } else { pr[0] = pr[1] = pr[2] = pr[3] = .25; }
        for (int j = 0; j < 5; j++) {
            double sum = 0;
            for (int k = 0; k < 4; k++) { if (j == k) continue; sum += pr[k] * T[k * 5 + j]; }
            if (j < 4) { FigPQ v; v.p = pr[j]; v.q = sum; fig_pq_set(E, j, x, v); }
            else fig_q4_set(E, x, sum);
        }
    }
    fig_build_flank_pq(E);
}
FIG_D void fig_compute_errprobs_only(FigEng &E) {
    int n = E.S->ncols;
    const double *T = E.M->T;
    for (int x = E.tid; x < n; x += E.nt) {
        double pr[4];
        for (int j = 0; j < 4; j++) pr[j] = fig_pq_get_p(E, j, x);
        for (int j = 0; j < 5; j++) {
            double sum = 0;
            for (int k = 0; k < 4; k++) { if (j == k) continue; sum += pr[k] * T[k * 5 + j]; }
            if (j < 4) fig_pq_set_q(E, j, x, sum); else fig_q4_set(E, x, sum);
        }
    }
    fig_build_flank_pq(E);
}

// computeSequence, Figbird.cpp:4417-4508.  All lanes; ends with a barrier.
FIG_D void fig_compute_sequence(FigEng &E, int check, int choice) {
    int n = E.S->ncols, cg = E.capG;
    const double *A = choice == 0 ? E.scr.cnt : E.scr.ncnt;
    for (int x = E.tid; x < n; x += E.nt) {
        double mx = 0; int mi = -1;
        for (int j = 0; j <= 4; j++) { double v = A[j * cg + x]; if (v > mx) { mx = v; mi = j; } }
        int coverage_flag = 1;
        if (check == 1) { int cv = (int)mx; E.scr.cov[x] = cv; if (cv <= 0) coverage_flag = 0; }
        unsigned char c = 4;
        if (((mi != -1) || (!check)) && coverage_flag) c = (mi >= 0 && mi <= 3) ? (unsigned char)mi : 4;
        E.scr.cons[x] = c;
    }
    if (E.tid == 0) E.S->cons_len = n < 0 ? 0 : n;
    FIG_SYNC();
}

FIG_D void fig_copy_str(FigEng &E, unsigned char *dst, int *dlen, const unsigned char *src, int slen) {
    for (int x = E.tid; x < slen; x += E.nt) dst[x] = src[x];
    if (E.tid == 0) *dlen = slen;
    FIG_SYNC();
}

// ---------------------------------------------------------------------------------------
// libstdc++ std::sort replica (introsort, threshold 16) on triples keyed by v[0], so that reads
// with equal positions end up in exactly the order the reference binary leaves them in
// (findOverlapUnmapped / findDiscontinous, Figbird.cpp:2956, 4635).  Serial (lane 0).
FIG_D bool fig_tlt(const FigTrip &a, const FigTrip &b) { return a.v[0] < b.v[0]; }
FIG_D void fig_tswap(FigTrip &a, FigTrip &b) { FigTrip t = a; a = b; b = t; }

FIG_D void fig_adjust_heap(FigTrip *first, int holeIndex, int len, FigTrip value) {
    int topIndex = holeIndex, secondChild = holeIndex;
    while (secondChild < (len - 1) / 2) {
        secondChild = 2 * (secondChild + 1);
        if (fig_tlt(first[secondChild], first[secondChild - 1])) secondChild--;
        first[holeIndex] = first[secondChild];
        holeIndex = secondChild;
    }
    if ((len & 1) == 0 && secondChild == (len - 2) / 2) {
        secondChild = 2 * (secondChild + 1);
        first[holeIndex] = first[secondChild - 1];
        holeIndex = secondChild - 1;
    }
    int parent = (holeIndex - 1) / 2;
    while (holeIndex > topIndex && fig_tlt(first[parent], value)) {
        first[holeIndex] = first[parent];
        holeIndex = parent;
        parent = (holeIndex - 1) / 2;
    }
    first[holeIndex] = value;
}
FIG_D void fig_heapsort(FigTrip *first, int n) {         // __partial_sort(first,last,last): make_heap + sort_heap
    if (n < 2) return;
    for (int parent = (n - 2) / 2;; parent--) { FigTrip v = first[parent]; fig_adjust_heap(first, parent, n, v); if (parent == 0) break; }
    for (int last = n; last > 1; last--) { FigTrip v = first[last - 1]; first[last - 1] = first[0]; fig_adjust_heap(first, 0, last - 1, v); }
}
FIG_D void fig_unguarded_linear_insert(FigTrip *a, int last) {
    FigTrip val = a[last];
    int next = last - 1;
    while (fig_tlt(val, a[next])) { a[last] = a[next]; last = next; next--; }
    a[last] = val;
}
FIG_D void fig_insertion_sort(FigTrip *a, int first, int last) {
    if (first == last) return;
    for (int i = first + 1; i != last; ++i) {
        if (fig_tlt(a[i], a[first])) { FigTrip val = a[i]; for (int k = i; k > first; k--) a[k] = a[k - 1]; a[first] = val; }
        else fig_unguarded_linear_insert(a, i);
    }
}
FIG_D void fig_std_sort(FigTrip *a, int n) {
    if (n <= 0) return;
    int lg = 0; { int t = n; while (t > 1) { t >>= 1; lg++; } }
    int stack_f[64], stack_l[64], stack_d[64], sp = 0;
    stack_f[0] = 0; stack_l[0] = n; stack_d[0] = lg * 2; sp = 1;
    while (sp > 0) {
        sp--;
        int first = stack_f[sp], last = stack_l[sp], depth = stack_d[sp];
        while (last - first > 16) {
            if (depth == 0) { fig_heapsort(a + first, last - first); break; }
            --depth;
            int mid = first + (last - first) / 2;
            {   // __move_median_to_first(first, first+1, mid, last-1)
                int A = first + 1, Bm = mid, C = last - 1;
                if (fig_tlt(a[A], a[Bm])) {
                    if (fig_tlt(a[Bm], a[C])) fig_tswap(a[first], a[Bm]);
                    else if (fig_tlt(a[A], a[C])) fig_tswap(a[first], a[C]);
                    else fig_tswap(a[first], a[A]);
                } else if (fig_tlt(a[A], a[C])) fig_tswap(a[first], a[A]);
                else if (fig_tlt(a[Bm], a[C])) fig_tswap(a[first], a[C]);
                else fig_tswap(a[first], a[Bm]);
            }
            int lo = first + 1, hi = last;
            while (true) {
                while (fig_tlt(a[lo], a[first])) ++lo;
                --hi;
                while (fig_tlt(a[first], a[hi])) --hi;
                if (!(lo < hi)) break;
                fig_tswap(a[lo], a[hi]);
                ++lo;
            }
            int cut = lo;
            // recurse on [cut,last) first (as the reference does), then loop on [first,cut)
            if (sp < 63) { stack_f[sp] = first; stack_l[sp] = cut; stack_d[sp] = depth; sp++; }
            first = cut;
            // NB: libstdc++ recurses into the right part and iterates on the left; the two
            // sub-ranges are disjoint, so processing order does not change the result.
        }
    }
    if (n > 16) { fig_insertion_sort(a, 0, 16); for (int i = 16; i != n; ++i) fig_unguarded_linear_insert(a, i); }
    else fig_insertion_sort(a, 0, n);
}

#include "fig_engine_core.h"

#endif
