// fig_abi.hip -- libfighip.so: the C ABI of include/figbird_hip.h on top of the gfx950 engine.
//
// Host side here = the "packer" role of Figbird.cpp's per-gap loop (Figbird.cpp:7329-7440):
// findFrac/alloc_arg (:6879-6906, :7393-7400), flank extraction, 2-bit read packing, cost
// sorting; then persistent-workgroup launches of fig_fill_kernel, one launch per LDS class.
// There is no CPU compute path in this library.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>
#include <chrono>
#include <thread>
#include <mutex>

#include "../../include/figbird_hip.h"
#include "fig_engine.h"
#include "fig_pack.h"

// ------------------------------------------------------------------------------------- kernel
// LDS carve-up (doubles first so everything stays 8-byte aligned), LDS_TAB only:
//   PQ[4][ncolE] (16 B each)  Q4[ncolE]  wbuf[nteams][Wcap]  | FigState  gs[capGl]  rb[FIG_MAX_READLEN+8]
#ifdef FIG_PROF
#define FIG_PROF_BEGIN() const unsigned long long _k0 = __builtin_readcyclecounter()
#define FIG_PROF_FLUSH() do { if (E.lane == 0) { atomicAdd(&B.counters[30], E.wait_cycles); atomicAdd(&B.counters[31], (unsigned long long)__builtin_readcyclecounter() - _k0); \
    for (int i = 0; i < 22; i++) if (E.prof[i]) atomicAdd(&B.counters[8 + i], E.prof[i]); \
    for (int i = 22; i < 40; i++) if (E.prof[i]) atomicAdd(&B.counters[32 + (i - 22)], E.prof[i]); } } while (0)
#else
#define FIG_PROF_BEGIN() ((void)0)
#define FIG_PROF_FLUSH() ((void)0)
#endif
// MLE-pass accounting (raw totals incl. discarded speculation; only their ratio is reported): [3] credited, [4] executed
#define FIG_FLUSH_MLE() do { if (E.mle_alg) atomicAdd(&B.counters[3], E.mle_alg); if (E.lane == 0 && E.mle_exec) atomicAdd(&B.counters[4], E.mle_exec); } while (0)
struct FigKernArgs { int capG, capGl, ncolE, Wcap, nteams, q_begin, q_end, qsel, tiles, tile_step, tile_cols, tiled_max, sh_on; };

FIG_D void fig_eng_init(FigEng &E, const FigDevModel &M, const FigDevBatch &B, const FigKernArgs &A, bool lds_tab, FigScr &work) {
    E.tid = threadIdx.x; E.nt = blockDim.x;
    E.lane = threadIdx.x & 63; E.wave = threadIdx.x >> 6; E.nw = (blockDim.x + 63) >> 6; E.wsz = 64;
    E.M = &M; E.B = &B;
    E.capG = A.capG; E.flops = 0; E.mle_alg = 0; E.mle_exec = 0; E.wait_cycles = 0;
    for (int i = 0; i < 40; i++) E.prof[i] = 0;
    E.ncolE = A.ncolE; E.xoff = M.L - 1; E.Wcap = A.Wcap; E.nteams = A.nteams; E.sh_on = A.sh_on;
    unsigned char *slab = B.scratch + (long long)blockIdx.x * B.scratch_stride;
    fig_scratch_layout(slab, B.capG, B.capR, B.capP, B.capC, B.capW, B.capE, &work);
    E.scr = work;
    long long off = 0;
    E.pq_lds = lds_tab; E.w_lds = lds_tab;
    E.tiles = A.tiles; E.tile_step = A.tile_step; E.tile_cols = A.tile_cols;
    if (lds_tab && A.tiles > 0) {
        // LDS-tiled class: LDS = {table tile [4][tile_cols] + Q4 tile, weight rows}; the full table, and the buffers of the
        // MLE pass (which runs its HBM-table form), stay in the scratch slab
        E.pq_lds = 0;
        E.off_pq = 0; E.off_q4 = 8 * A.tile_cols; E.off_w = 9 * A.tile_cols;
        off = 9LL * A.tile_cols + (long long)A.nteams * A.Wcap;
        { const long long mle = FIG_TILED_MLE_DOUBLES(A.ncolE, (int)E.nw); if (mle > off && mle <= A.tiled_max) off = mle; }   // as fig_pack sized it
        E.pq = E.scr.pqg; E.q4 = E.scr.q4g; E.wbuf = E.scr.wg;
    } else if (lds_tab) {
        E.off_pq = 0; E.off_q4 = 8 * A.ncolE; E.off_w = 9 * A.ncolE;
        off = 9LL * A.ncolE + (long long)A.nteams * A.Wcap;
        E.pq = (FigPQ *)(fig_lds + E.off_pq); E.q4 = fig_lds + E.off_q4; E.wbuf = fig_lds + E.off_w;
    } else {
        E.off_pq = E.off_q4 = E.off_w = 0;
        E.pq = E.scr.pqg; E.q4 = E.scr.q4g; E.wbuf = E.scr.wg;
    }
    E.S = (FigState *)(fig_lds + off); E.lds_tw = (int)off;
    unsigned char *bp = (unsigned char *)(E.S + 1);
    E.gs = bp; bp += ((A.capGl + 7) & ~7);
    E.rb = bp; bp += ((FIG_MAX_READLEN + 8 + 15) & ~15);
    E.plb = (uint32_t *)bp; E.off_plb = (int)((double *)bp - fig_lds);
    E.kt_fwd = M.ome; E.kt_rev = M.ome + 2 * M.L; E.mt_fwd = M.ome + 4 * M.L; E.mt_rev = M.ome + 6 * M.L;
}

FIG_D void fig_persist_of(const FigDevBatch &B, const FigDevGap &g, FigPersist &P) {
    fig_persist_layout(B.persist + g.persistOff, g.capGg, g.nU, g.nP, g.rangeCap, g.nslots, sizeof(FigState), &P);
}

// ---- sequential mode: whole gaps, one workgroup each (FIG_SCHED=seq)
template <bool LDS_TAB, int NT>
__global__ void __launch_bounds__(NT, (NT <= 256 ? 2 : 1)) fig_fill_kernel(FigDevModel M, FigDevBatch B, FigKernArgs A) {
    FigEng E; FigScr work;
    fig_eng_init(E, M, B, A, LDS_TAB, work);
    if (blockIdx.x == 0 && threadIdx.x == 0) B.queue_head[A.qsel ^ 1] = 0;      // the next launch of this lane pops from the other head
    FIG_PROF_BEGIN();
    while (true) {
        if (E.tid == 0) E.S->bc_i = atomicAdd(B.queue_head + A.qsel, 1);
        __syncthreads();
        int qi = A.q_begin + E.S->bc_i;
        __syncthreads();
        if (qi >= A.q_end) break;
        E.g = &B.gaps[B.order[qi]];
        FigPersist P; fig_persist_of(B, *E.g, P);
        fig_bind(E, work, P, FIG_BIND_SEQ, 0, E.g->capGg);
        fig_fill_gap<LDS_TAB>(E);
    }
    FIG_PROF_FLUSH();
    if (E.flops) atomicAdd(&B.counters[1], E.flops);
    FIG_FLUSH_MLE();
}

// ---- candidate-parallel mode, kernel 1: setup + analyzeGap + checkGapReads per gap; skipped gaps finish here
template <bool LDS_TAB, int NT>
__global__ void __launch_bounds__(NT, (NT <= 256 ? 2 : 1)) fig_begin_kernel(FigDevModel M, FigDevBatch B, FigKernArgs A) {
    FigEng E; FigScr work;
    fig_eng_init(E, M, B, A, LDS_TAB, work);
    if (blockIdx.x == 0 && threadIdx.x == 0) B.queue_head[A.qsel ^ 1] = 0;      // the next launch of this lane pops from the other head
    while (true) {
        if (E.tid == 0) E.S->bc_i = atomicAdd(B.queue_head + A.qsel, 1);
        __syncthreads();
        int qi = A.q_begin + E.S->bc_i;
        __syncthreads();
        if (qi >= A.q_end) break;
        int gi = B.order[qi];
        E.g = &B.gaps[gi];
        FigPersist P; fig_persist_of(B, *E.g, P);
        fig_bind(E, work, P, FIG_BIND_SEQ, 0, E.g->capGg);
        fig_gap_begin<LDS_TAB>(E);
        if (!E.S->L.inr) {                               // not filled (:6227): one pass without EM, then output
            while (!E.S->L.done) { fig_eval_candidate<LDS_TAB>(E); fig_loop_step(E); }
            fig_gap_end<LDS_TAB>(E);
            if (E.tid == 0) { B.gapctl[gi * 4] = 0; }
        } else {
            fig_state_save(E, P);
            if (E.tid == 0) { B.gapctl[gi * 4] = E.S->L.done ? 2 : 1; B.gapctl[gi * 4 + 1] = E.S->L.j; B.gapctl[gi * 4 + 2] = E.S->L.range; }
        }
        __syncthreads();
    }
    if (E.flops) atomicAdd(&B.counters[1], E.flops);
    FIG_FLUSH_MLE();
}

// ---- pre-pass (partial mode): per gap, does its candidate loop get to Figbird.cpp:6317?  -> gapctl[gi*4+3]
template <bool LDS_TAB, int NT>
__global__ void __launch_bounds__(NT, (NT <= 256 ? 2 : 1)) fig_probe_kernel(FigDevModel M, FigDevBatch B, FigKernArgs A) {
    FigEng E; FigScr work;
    fig_eng_init(E, M, B, A, LDS_TAB, work);
    if (blockIdx.x == 0 && threadIdx.x == 0) B.queue_head[A.qsel ^ 1] = 0;      // the next launch of this lane pops from the other head
    while (true) {
        if (E.tid == 0) E.S->bc_i = atomicAdd(B.queue_head + A.qsel, 1);
        __syncthreads();
        int qi = A.q_begin + E.S->bc_i;
        __syncthreads();
        if (qi >= A.q_end) break;
        int gi = B.order[qi];
        E.g = &B.gaps[gi];
        FigPersist P; fig_persist_of(B, *E.g, P);
        fig_bind(E, work, P, FIG_BIND_SEQ, 0, E.g->capGg);
        const int reach = fig_gap_probe<LDS_TAB>(E);
        if (E.tid == 0) B.gapctl[gi * 4 + 3] = reach;
        __syncthreads();
    }
}

// ---- kernel 2: speculative candidate evaluations; items = {gap, candidate index j, slot, -}
template <bool LDS_TAB, int NT>
__global__ void __launch_bounds__(NT, (NT <= 256 ? 2 : 1)) fig_eval_kernel(FigDevModel M, FigDevBatch B, FigKernArgs A, const int4 *items, int n_items) {
    FigEng E; FigScr work;
    fig_eng_init(E, M, B, A, LDS_TAB, work);
    if (blockIdx.x == 0 && threadIdx.x == 0) B.queue_head[A.qsel ^ 1] = 0;      // the next launch of this lane pops from the other head
    FIG_PROF_BEGIN();
    while (true) {
        if (E.tid == 0) E.S->bc_i = atomicAdd(B.queue_head + A.qsel, 1);
        __syncthreads();
        int qi = E.S->bc_i;
        __syncthreads();
        if (qi >= n_items) break;
        int4 it = items[qi];
        E.g = &B.gaps[it.x];
        FigPersist P; fig_persist_of(B, *E.g, P);
        fig_spec_eval<LDS_TAB>(E, work, P, it.y, it.z, E.g->capGg);
        __syncthreads();
    }
    FIG_PROF_FLUSH();
    if (E.flops) atomicAdd(&B.counters[1], E.flops);
    FIG_FLUSH_MLE();
}

// ---- kernel 3: replay the bookkeeping of the speculated candidates in order; entries = {gap, n slots, -, -}
__global__ void __launch_bounds__(64) fig_replay_kernel(FigDevModel M, FigDevBatch B, const int4 *entries, int n) {
    if ((int)blockIdx.x >= n) return;
    FigEng E; FigScr work;
    memset(&work, 0, sizeof(work));
    E.tid = threadIdx.x; E.nt = blockDim.x; E.lane = threadIdx.x & 63; E.wave = 0; E.nw = 1; E.wsz = 64;
    E.M = &M; E.B = &B; E.capG = 0; E.flops = 0; E.mle_alg = 0; E.mle_exec = 0; E.wait_cycles = 0; E.ncolE = 0; E.xoff = 0; E.Wcap = 0; E.nteams = 1; E.sh_on = 0;
    E.pq = nullptr; E.q4 = nullptr; E.wbuf = nullptr; E.gs = nullptr; E.rb = nullptr; E.plb = nullptr; E.off_plb = 0; E.pq_lds = 0; E.w_lds = 0;
    E.off_pq = E.off_q4 = E.off_w = 0; E.tiles = E.tile_step = E.tile_cols = 0; E.lds_tw = 0;
    E.S = (FigState *)fig_lds;
    int4 en = entries[blockIdx.x];
    E.g = &B.gaps[en.x];
    FigPersist P; fig_persist_of(B, *E.g, P);
    fig_spec_replay(E, work, P, en.y, E.g->capGg);
    if (E.tid == 0) { B.gapctl[en.x * 4] = E.S->L.done ? 2 : 1; B.gapctl[en.x * 4 + 1] = E.S->L.j; B.gapctl[en.x * 4 + 2] = E.S->L.range; }
}

// ---- kernel 4: fallbacks + finalize + output for the gaps whose loop is done; list = gap ids
template <bool LDS_TAB, int NT>
__global__ void __launch_bounds__(NT, (NT <= 256 ? 2 : 1)) fig_end_kernel(FigDevModel M, FigDevBatch B, FigKernArgs A, const int *list, int n) {
    FigEng E; FigScr work;
    fig_eng_init(E, M, B, A, LDS_TAB, work);
    if (blockIdx.x == 0 && threadIdx.x == 0) B.queue_head[A.qsel ^ 1] = 0;      // the next launch of this lane pops from the other head
    while (true) {
        if (E.tid == 0) E.S->bc_i = atomicAdd(B.queue_head + A.qsel, 1);
        __syncthreads();
        int qi = E.S->bc_i;
        __syncthreads();
        if (qi >= n) break;
        int gi = list[qi];
        E.g = &B.gaps[gi];
        FigPersist P; fig_persist_of(B, *E.g, P);
        fig_bind(E, work, P, FIG_BIND_SEQ, 0, E.g->capGg);
        fig_state_load(E, P);
        fig_gap_end<LDS_TAB>(E);
        if (E.tid == 0) B.gapctl[gi * 4] = 0;
        __syncthreads();
    }
    if (E.flops) atomicAdd(&B.counters[1], E.flops);
    FIG_FLUSH_MLE();
}

// ---- upload-time kernel: the operand-select stream of the shared-factor E-step (fig_engine_shared.h).  Entry (read r, chain
// step j) = 0x1000 | 2 * (4 * reverse + base_j): the value M0 takes so that `v_mul_f64 p, v[F0:F0+1], p` in VGPR-index mode
// multiplies by the factor f[orientation][base].  Reads the fast path cannot take (an N base, a length other than L) and the
// empty slots of a gap's last chunk select the constant 1.0.  Layout per gap: [chunk of 32 reads][step][32 reads] 16-bit
// entries, two per dword.  One block per gap.
__global__ void __launch_bounds__(256) fig_stream_kernel(FigDevModel M, FigDevBatch B, uint32_t *out) {
    for (long long gi = blockIdx.x; gi < B.n_gaps; gi += gridDim.x) {
        const FigDevGap &g = B.gaps[gi];
        const int L = M.L;
        const long long n = (long long)((g.nU + FIG_SH_C - 1) / FIG_SH_C) * L * (FIG_SH_C / 2);
        for (long long i = threadIdx.x; i < n; i += blockDim.x) {
            const int sp = (int)(i % (FIG_SH_C / 2));
            const long long cj = i / (FIG_SH_C / 2);
            const int j = (int)(cj % L);
            uint32_t v = 0;
            for (int h = 0; h < 2; h++) {
                const long long r = (cj / L) * FIG_SH_C + 2 * sp + h;
                unsigned e = 0x1000u | FIG_SH_ONE;
                if (r < g.nU) {
                    const long long idx = g.uBase + r;
                    const int len = B.u.len[idx], aux = B.u.aux[idx];
                    if (len == L && !(aux & 2)) {
                        const uint32_t w = B.packed[B.u.woff[idx] + (j >> 4)];
                        e = 0x1000u | (2u * (4u * (unsigned)(aux & 1) + ((w >> ((j & 15) * 2)) & 3u)));
                    }
                }
                v |= e << (16 * h);
            }
            out[g.streamOff + i] = v;
        }
    }
}

// ------------------------------------------------------------------------------------- context
#define FIG_HIP(call) do { hipError_t _e = (call); if (_e != hipSuccess) { ctx->last_hip = (int)_e; return FIG_EHIP; } } while (0)

struct DevBuf {
    void *p = nullptr; size_t n = 0;
};

// Per-class scheduling lane: the classes of a batch run concurrently, each on its own stream with its own work
// queue head, scratch slabs and item buffers, so that the tail of one class's round is filled by the other
// classes' workgroups.
struct FigLane { hipStream_t stream = nullptr; hipEvent_t done = nullptr; int32_t *queue_head = nullptr; uint8_t *scratch = nullptr; int *d_items = nullptr, *d_entries = nullptr; size_t cap = 0;
                 int32_t *h_ctl = nullptr; int *h_items = nullptr, *h_entries = nullptr; int qsel = 0; };   // h_*: pinned, so the copies run on the DMA engines and never wait for a CU

struct fig_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int n_cu = 0;
    int last_hip = 0;
    bool have_model = false;
    FigDevModel dm;
    std::vector<double> h_e, h_ome, h_m3, h_insd, h_qtab;
    DevBuf d_model_tabs;
    fig_model hm;
    // resident batch
    bool have_batch = false;
    FigDevBatch db;
    std::vector<FigDevGap> h_gaps;
    std::vector<int32_t> h_order;
    struct Cls { FigLaunchClass c; int blocks; int capacity; };
    std::vector<Cls> classes;
    std::vector<DevBuf> bufs;
    int64_t n_gaps = 0, n_ureads = 0, n_preads = 0, str_total = 0;
    std::vector<int64_t> h_str_off;
    int nslots = 32;
    std::vector<FigLane> lanes;
    fig_stats stats;
    int sh_on = FIG_SH_SC;            // FIG_ESTEP / FIG_SH_CHUNKS, read once at fig_ctx_create
    std::vector<uint8_t> h_ot;        // host copy of db.ot_preset
    uint8_t *d_ot = nullptr;
};

static int dev_alloc(fig_ctx *ctx, size_t bytes, void **out) {
    DevBuf b;
    if (bytes == 0) bytes = 8;
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess) { ctx->last_hip = (int)e; return FIG_ENOMEM; }
    b.n = bytes;
    ctx->bufs.push_back(b);
    *out = b.p;
    return FIG_OK;
}

template <typename T>
static int dev_upload(fig_ctx *ctx, const std::vector<T> &v, const T **out) {
    void *p = nullptr;
    int rc = dev_alloc(ctx, v.size() * sizeof(T), &p);
    if (rc) return rc;
    if (!v.empty()) {
        hipError_t e = hipMemcpyAsync(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) { ctx->last_hip = (int)e; return FIG_EHIP; }
    }
    *out = (const T *)p;
    return FIG_OK;
}

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

extern "C" int fig_ctx_create(int device_ordinal, fig_ctx **out) {
    if (!out) return FIG_EINVAL;
    *out = nullptr;
    // A fill runs up to nine streams (one per class lane + the context's own); the HIP runtime maps streams onto
    // GPU_MAX_HW_QUEUES hardware queues (default 4), and two lanes' persistent kernels on one queue serialise.  Asking for 8
    // is worth 1 % of the bench step.  Only effective when this is the process's first HIP call (figfill); a host that
    // initialises HIP earlier sets the variable itself (bench.py and figfill_mp do).
    static std::once_flag env_once;   // contexts may be created from several host threads (figfill with FIGFILL_DEVICES sets it in main() already)
    std::call_once(env_once, [] { setenv("GPU_MAX_HW_QUEUES", "8", 0); });
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return FIG_ENODEV;
    if (device_ordinal < 0 || device_ordinal >= n) return FIG_ENODEV;
    if (hipSetDevice(device_ordinal) != hipSuccess) return FIG_ENODEV;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess) return FIG_ENODEV;
    fig_ctx *ctx = new (std::nothrow) fig_ctx();
    if (!ctx) return FIG_ENOMEM;
    ctx->device = device_ordinal;
    ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hipStreamCreate(&ctx->stream) != hipSuccess || hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
        delete ctx;
        return FIG_EHIP;
    }
    memset(&ctx->stats, 0, sizeof(ctx->stats));
    { const char *ev = getenv("FIG_ESTEP"); const char *sc = getenv("FIG_SH_CHUNKS");
      ctx->sh_on = (ev && !strcmp(ev, "pair")) ? 0 : (sc ? std::max(1, std::min(FIG_SH_SC, atoi(sc))) : FIG_SH_SC); }
    *out = ctx;
    return FIG_OK;
}

static void free_batch(fig_ctx *ctx) {
    for (auto &b : ctx->bufs) if (b.p) hipFree(b.p);
    ctx->bufs.clear();
    ctx->d_ot = nullptr;
    ctx->have_batch = false;
    ctx->classes.clear();
    for (auto &l : ctx->lanes) { if (l.stream) hipStreamDestroy(l.stream); if (l.done) hipEventDestroy(l.done); if (l.h_ctl) hipHostFree(l.h_ctl); if (l.h_items) hipHostFree(l.h_items); if (l.h_entries) hipHostFree(l.h_entries); }
    ctx->lanes.clear();
}

extern "C" void fig_batch_free(fig_ctx *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    free_batch(ctx);
}

extern "C" void fig_ctx_destroy(fig_ctx *ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    free_batch(ctx);
    if (ctx->d_model_tabs.p) hipFree(ctx->d_model_tabs.p);
    if (ctx->ev0) hipEventDestroy(ctx->ev0);
    if (ctx->ev1) hipEventDestroy(ctx->ev1);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int fig_ctx_set_model(fig_ctx *ctx, const fig_model *m) {
    if (!ctx || !m || !m->error_pos_dist || !m->in_pos_dist || !m->del_pos_dist || !m->insert_len_dist_smoothed) return FIG_EINVAL;
    if (m->max_read_length <= 0 || m->max_read_length > FIG_MAX_READLEN) return FIG_EUNSUP;
    if (m->max_insert_size <= 0) return FIG_EINVAL;
    if (m->partial_flag && m->unmapped_flag) return FIG_EUNSUP;      // the driver never sets both (RunFigbird.sh:211-216)
    if (m->insert_threshold_min < 0 || m->insert_threshold_max >= m->max_insert_size + 1) return FIG_EINVAL;
    hipSetDevice(ctx->device);
    free_batch(ctx);                  // a resident batch was packed (classes, capacities, candidate ranges) under the previous model
    int L = m->max_read_length;
    ctx->hm = *m;
    ctx->h_e.assign(m->error_pos_dist, m->error_pos_dist + L);
    ctx->h_ome.resize(L); ctx->h_m3.resize(L);
    for (int k = 0; k < L; k++) {
        volatile double a = 1 - m->error_pos_dist[k];                 // (1-errorPosDist[k])               Figbird.cpp:3160
        ctx->h_ome[k] = a;
        volatile double b = 1 - m->error_pos_dist[k] - m->in_pos_dist[k] - m->del_pos_dist[k];   // Figbird.cpp:3400
        ctx->h_m3[k] = b;
    }
    ctx->h_insd.assign(m->insert_len_dist_smoothed, m->insert_len_dist_smoothed + m->max_insert_size);
    ctx->h_insd.push_back(0.0);                                       // insertThresholdMax may equal maxInsertSize (:7194)
    ctx->h_qtab.resize(256);
    for (int c = 0; c < 256; c++) { int Q = c - 33; ctx->h_qtab[c] = pow(10, -Q / 10.0); }   // qualityFilter, :1791-1792
    // pair tables for scalar loads: kt = {1-e[k], e[k]}, mt = {1-e-ins-del, e[k]}; the reversed copies serve
    // reverse-strand reads (readIndex = len-1-j, Figbird.cpp:3569-3576) as rev[(L-len)+j]
    std::vector<double> pairs((size_t)8 * L);
    for (int k = 0; k < L; k++) {
        pairs[2 * k] = ctx->h_ome[k]; pairs[2 * k + 1] = ctx->h_e[k];
        pairs[2 * L + 2 * k] = ctx->h_ome[L - 1 - k]; pairs[2 * L + 2 * k + 1] = ctx->h_e[L - 1 - k];
        pairs[4 * L + 2 * k] = ctx->h_m3[k]; pairs[4 * L + 2 * k + 1] = ctx->h_e[k];
        pairs[6 * L + 2 * k] = ctx->h_m3[L - 1 - k]; pairs[6 * L + 2 * k + 1] = ctx->h_e[L - 1 - k];
    }
    std::vector<double> all;
    size_t o_e = 0; all.insert(all.end(), ctx->h_e.begin(), ctx->h_e.end()); while (all.size() % 8) all.push_back(0);
    size_t o_pairs = all.size(); all.insert(all.end(), pairs.begin(), pairs.end()); while (all.size() % 8) all.push_back(0);
    size_t o_m3 = all.size(); all.insert(all.end(), ctx->h_m3.begin(), ctx->h_m3.end()); while (all.size() % 8) all.push_back(0);
    size_t o_insd = all.size(); all.insert(all.end(), ctx->h_insd.begin(), ctx->h_insd.end()); while (all.size() % 8) all.push_back(0);
    size_t o_q = all.size(); all.insert(all.end(), ctx->h_qtab.begin(), ctx->h_qtab.end());
    size_t o_ome = all.size(); all.insert(all.end(), ctx->h_ome.begin(), ctx->h_ome.end());
    size_t nd = all.size();
    if (ctx->d_model_tabs.p) { hipFree(ctx->d_model_tabs.p); ctx->d_model_tabs.p = nullptr; }
    if (hipMalloc(&ctx->d_model_tabs.p, nd * sizeof(double)) != hipSuccess) return FIG_ENOMEM;
    double *d = (double *)ctx->d_model_tabs.p;
    if (hipMemcpy(d, all.data(), nd * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return FIG_EHIP;
    FigDevModel &dm = ctx->dm;
    dm.L = L; dm.Tmin = m->insert_threshold_min; dm.Tmax = m->insert_threshold_max; dm.cutoff = m->gap_prob_cutoff;
    dm.partial_flag = m->partial_flag; dm.unmapped = m->unmapped_flag; dm.script_itr = m->script_itr; dm.D = m->max_distance;
    dm.read_length = m->read_length; dm.neg_overlap = m->neg_overlap; dm.partial_len = m->partial_len; dm.unm_limit = m->unm_limit;
    dm.max_insert = m->max_insert_size;
    for (int i = 0; i < 25; i++) dm.T[i] = m->error_type_probs[i];
    dm.fmm_up = fig_model_fmm(m);
    dm.e = d + o_e; dm.ome = d + o_pairs; dm.m3 = d + o_m3; dm.insd = d + o_insd; dm.qtab = d + o_q;
    dm.ome1 = d + o_ome;
    ctx->have_model = true;
    return FIG_OK;
}

extern "C" int64_t fig_results_capacity(const fig_model *m, const fig_gap_batch *b) {
    if (!m || !b) return FIG_EINVAL;
    return fig_pack_results_capacity(m, b);
}

static hipError_t launch_any(fig_ctx *ctx, const fig_ctx::Cls &c, const FigDevBatch &db, hipStream_t stream, int kind, int blocks, const void *list, int n, int qsel = 0);

extern "C" int fig_batch_upload(fig_ctx *ctx, const fig_gap_batch *b) {
    if (!ctx || !b) return FIG_EINVAL;
    if (!ctx->have_model) return FIG_EINVAL;
    hipSetDevice(ctx->device);
    free_batch(ctx);
    const fig_model *m = &ctx->hm;
    hipEventRecord(ctx->ev0, ctx->stream);
    FigPacked K;
    int prc = fig_pack(m, b, sizeof(FigState), K);
    if (prc) return prc;
    int64_t ng = K.n_gaps;
    ctx->h_gaps = K.gaps; ctx->h_order = K.order; ctx->h_str_off = K.str_off;
    ctx->str_total = K.str_total; ctx->n_gaps = ng;
    ctx->n_ureads = (int64_t)K.u_pos.size(); ctx->n_preads = (int64_t)K.p_pos.size();
    ctx->classes.clear();
    int max_blocks = 1;
    for (const FigLaunchClass &lc : K.classes) {
        fig_ctx::Cls c;
        c.c = lc;
        int per_cu = (int)std::max<size_t>(1, std::min<size_t>((160 * 1024) / std::max<size_t>(lc.lds, 1), (size_t)(2048 / lc.nt)));
        per_cu = std::min(per_cu, 8);
        c.capacity = ctx->n_cu * per_cu;
        c.blocks = std::min<int>(lc.q_end - lc.q_begin, c.capacity);
        max_blocks = std::max(max_blocks, c.capacity);
        if (getenv("FIG_SCHED_LOG")) fprintf(stderr, "[figsched] class: capG=%d capGl=%d ncolE=%d Wcap=%d nt=%d nteams=%d lds_tab=%d tiles=%d tile_cols=%d lds=%zu gaps=%d per_cu=%d blocks=%d (FigState %zu B)\n",
                                             lc.capG, lc.capGl, lc.ncolE, lc.Wcap, lc.nt, lc.nteams, (int)lc.lds_tab, lc.tiles, lc.tile_cols, lc.lds, lc.q_end - lc.q_begin, per_cu, c.blocks, sizeof(FigState));
        ctx->classes.push_back(c);
    }
    int capG_s = K.capG, capR = K.capR, capP = K.capP, capC = K.capC;
    int64_t str_total = K.str_total;
    int64_t stride = fig_scratch_layout(nullptr, capG_s, capR, capP, capC, K.capW, K.capE, nullptr);
    stride = (stride + 255) & ~255LL;
    std::vector<FigDevGap> &gaps = K.gaps; std::vector<int32_t> &order = K.order;
    std::vector<uint32_t> &packed = K.packed; std::vector<uint8_t> &qual = K.qual, &flank = K.flank;
    std::vector<int32_t> &u_pos = K.u_pos, &u_aux = K.u_aux, &u_len = K.u_len, &p_pos = K.p_pos, &p_aux = K.p_aux, &p_clip = K.p_clip, &p_ref = K.p_ref, &p_len = K.p_len;
    std::vector<int64_t> &u_woff = K.u_woff, &p_woff = K.p_woff, &p_qoff = K.p_qoff;

    // ---- upload
    FigDevBatch &db = ctx->db;
    memset(&db, 0, sizeof(db));
    db.n_gaps = ng;
    int rc;
#define UP(vec, field) do { rc = dev_upload(ctx, vec, &db.field); if (rc) return rc; } while (0)
    UP(gaps, gaps); UP(order, order); UP(packed, packed); UP(qual, qual); UP(flank, flank);
    UP(u_pos, u.pos); UP(u_aux, u.aux); UP(u_len, u.len); UP(u_woff, u.woff);
    UP(p_pos, p.pos); UP(p_aux, p.aux); UP(p_clip, p.clip); UP(p_ref, p.refpos); UP(p_len, p.len); UP(p_woff, p.woff); UP(p_qoff, p.qoff);
#undef UP
    db.u.clip = nullptr; db.u.refpos = nullptr; db.u.qoff = nullptr;
    void *p;
    if (m->unmapped_flag) {
        if ((rc = dev_alloc(ctx, (size_t)K.stream_total * 4, &p))) return rc;
        db.ustream = (const uint32_t *)p;
        // every entry starts as "select the constant 1.0" in both halves: a slot the kernel does not write never switches indexing off
        FIG_HIP(hipMemsetD32Async((hipDeviceptr_t)p, (int)(((0x1000u | FIG_SH_ONE) << 16) | (0x1000u | FIG_SH_ONE)), (size_t)K.stream_total, ctx->stream));
        if (ng > 0) {
            hipLaunchKernelGGL(fig_stream_kernel, dim3((unsigned)std::min<int64_t>(ng, 4096)), dim3(256), 0, ctx->stream, ctx->dm, db, (uint32_t *)p);
            FIG_HIP(hipGetLastError());
        }
    }
    if ((rc = dev_alloc(ctx, (size_t)ng * 4, &p))) return rc; db.filled_len = (int32_t *)p;
    if ((rc = dev_alloc(ctx, (size_t)ng * 4, &p))) return rc; db.gaptofill = (int32_t *)p;
    if ((rc = dev_alloc(ctx, (size_t)str_total, &p))) return rc; db.str = (char *)p;
    if ((rc = dev_alloc(ctx, 64 * (ctx->classes.size() + 1), &p))) return rc; db.queue_head = (int32_t *)p;
    if ((rc = dev_alloc(ctx, 512, &p))) return rc; db.counters = (unsigned long long *)p;
    size_t total_blocks = 0;
    for (const fig_ctx::Cls &c : ctx->classes) total_blocks += (size_t)c.capacity;
    (void)max_blocks;
    if ((rc = dev_alloc(ctx, (size_t)stride * std::max<size_t>(total_blocks, 1), &p))) return rc; db.scratch = (uint8_t *)p;
    if ((rc = dev_alloc(ctx, (size_t)K.persist_total + 256, &p))) return rc; db.persist = (uint8_t *)p;
    if ((rc = dev_alloc(ctx, (size_t)std::max<int64_t>(ng, 1) * 16, &p))) return rc; db.gapctl = (int32_t *)p;
    if ((rc = dev_alloc(ctx, (size_t)std::max<int64_t>(ng, 1), &p))) return rc; ctx->d_ot = (uint8_t *)p; db.ot_preset = ctx->d_ot;
    ctx->h_ot = K.ot_preset;
    FIG_HIP(hipMemcpyAsync(ctx->d_ot, ctx->h_ot.data(), (size_t)std::max<int64_t>(ng, 1), hipMemcpyHostToDevice, ctx->stream));
    ctx->nslots = K.nslots;
    {   size_t blk = 0;
        for (size_t ci = 0; ci < ctx->classes.size(); ci++) {
            const fig_ctx::Cls &c = ctx->classes[ci];
            ctx->lanes.push_back(FigLane());          // pushed first: free_batch reclaims whatever was created when a later step fails
            FigLane &l = ctx->lanes.back();
            if (hipStreamCreate(&l.stream) != hipSuccess || hipEventCreateWithFlags(&l.done, hipEventDisableTiming) != hipSuccess) return FIG_EHIP;
            l.queue_head = db.queue_head + 16 * (ci + 1);
            l.scratch = db.scratch + (size_t)stride * blk; blk += (size_t)c.capacity;
            l.cap = (size_t)std::max(c.c.q_end - c.c.q_begin, 1) * 4 * (size_t)(K.nslots + 1);
            if ((rc = dev_alloc(ctx, l.cap * 4, &p))) return rc; l.d_items = (int *)p;
            if ((rc = dev_alloc(ctx, l.cap * 4, &p))) return rc; l.d_entries = (int *)p;
            if (hipHostMalloc((void **)&l.h_ctl, (size_t)std::max<int64_t>(ng, 1) * 16, hipHostMallocDefault) != hipSuccess) return FIG_ENOMEM;
            if (hipHostMalloc((void **)&l.h_items, l.cap * 4, hipHostMallocDefault) != hipSuccess) return FIG_ENOMEM;
            if (hipHostMalloc((void **)&l.h_entries, l.cap * 4, hipHostMallocDefault) != hipSuccess) return FIG_ENOMEM;
        }
    }
    db.scratch_stride = stride;
    db.capG = capG_s; db.capR = capR; db.capP = capP; db.capC = capC; db.capW = K.capW; db.capE = K.capE;
    db.n_ureads = ctx->n_ureads;
    FIG_HIP(hipMemsetAsync(db.counters, 0, 512, ctx->stream));
    FIG_HIP(hipMemsetAsync(db.queue_head, 0, 64 * (ctx->classes.size() + 1), ctx->stream));
    hipEventRecord(ctx->ev1, ctx->stream);
    FIG_HIP(hipStreamSynchronize(ctx->stream));
    float ms = 0; hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    ctx->stats.h2d_ms = ms;
    ctx->stats.packed_bytes = K.packed_bytes();
    ctx->have_batch = true;
    if (!K.ot_given && m->partial_flag && ng > 0) {
        // no carry given: the batch is ONE worker process of the reference taking its gaps in batch order -- measure which
        // gaps get to Figbird.cpp:6317 and hand every gap the prefix-OR of its predecessors
        std::vector<uint8_t> reach((size_t)ng, 0), preset((size_t)ng, 0);
        if ((rc = fig_batch_probe_reach(ctx, reach.data()))) { free_batch(ctx); return rc; }
        std::vector<int64_t> ids((size_t)ng);
        for (int64_t g = 0; g < ng; g++) ids[(size_t)g] = g;
        fig_ot_carry(ids, reach.data(), preset.data());
        if ((rc = fig_batch_set_ot_preset(ctx, preset.data()))) { free_batch(ctx); return rc; }
    }
    return FIG_OK;
}

// Per gap of the resident batch: 1 iff the gap's candidate loop gets to Figbird.cpp:6317 (fig_engine_sched.h: fig_gap_probe).
// Partial mode: one cheap launch per class (setup + the first initialize(), no EM); unmapped mode: all 0 (overlap_threshold is
// never read there, see fig_gap_probe).
extern "C" int fig_batch_probe_reach(fig_ctx *ctx, uint8_t *reach) {
    if (!ctx || !reach || !ctx->have_batch) return FIG_EINVAL;
    hipSetDevice(ctx->device);
    const int64_t ng = ctx->n_gaps;
    memset(reach, 0, (size_t)ng);
    if (!ctx->hm.partial_flag || ng == 0) return FIG_OK;
    FigDevBatch db = ctx->db;
    db.dbg_n_cand = nullptr; db.dbg_cand_i = nullptr; db.dbg_cand_lik = nullptr; db.dbg_max_cand = 0; db.dbg_n_place = nullptr;
    db.draw_pos = db.draw_isz = db.draw_len = nullptr;
    db.dbg_counts = db.dbg_read_maxlv = nullptr; db.dbg_plane_cols = db.dbg_plane_reads = 0;
    FIG_HIP(hipMemsetAsync(db.gapctl, 0, (size_t)ng * 16, ctx->stream));
    for (const fig_ctx::Cls &c : ctx->classes) {
        FIG_HIP(hipMemsetAsync(db.queue_head, 0, 8, ctx->stream));
        FIG_HIP(launch_any(ctx, c, db, ctx->stream, 4, c.blocks, nullptr, 0));
    }
    std::vector<int32_t> ctl((size_t)ng * 4);
    FIG_HIP(hipMemcpyAsync(ctl.data(), db.gapctl, (size_t)ng * 16, hipMemcpyDeviceToHost, ctx->stream));
    FIG_HIP(hipStreamSynchronize(ctx->stream));
    for (int64_t g = 0; g < ng; g++) reach[g] = ctl[(size_t)g * 4 + 3] ? 1 : 0;
    return FIG_OK;
}

// Replaces fig_gap_batch::gap_ot_preset of the resident batch (a caller that fills a shard of a run: probe every shard,
// exchange the bits, carry them along the reference's worker processes -- fig_gaprules.h -- and set the result here).
extern "C" int fig_batch_set_ot_preset(fig_ctx *ctx, const uint8_t *preset) {
    if (!ctx || !preset || !ctx->have_batch) return FIG_EINVAL;
    hipSetDevice(ctx->device);
    const int64_t ng = ctx->n_gaps;
    for (int64_t g = 0; g < ng; g++) ctx->h_ot[(size_t)g] = preset[g] ? 1 : 0;
    if (ng > 0) { FIG_HIP(hipMemcpy(ctx->d_ot, ctx->h_ot.data(), (size_t)ng, hipMemcpyHostToDevice)); }
    return FIG_OK;
}

static FigKernArgs kargs_of(const fig_ctx::Cls &c) {
    FigKernArgs A;
    A.capG = c.c.capG; A.capGl = c.c.capGl; A.ncolE = c.c.ncolE; A.Wcap = c.c.Wcap; A.nteams = c.c.nteams;
    A.q_begin = c.c.q_begin; A.q_end = c.c.q_end; A.qsel = 0;
    A.tiles = c.c.tiles; A.tile_step = c.c.tile_step; A.tile_cols = c.c.tile_cols; A.tiled_max = c.c.tiled_max;
    // FIG_ESTEP=pair: the pair-chain E-step everywhere (A/B runs, tests); FIG_SH_CHUNKS=<1..4>: chunks per super-chunk of the
    // shared-factor E-step (fig_engine_shared.h; default 4: bench step 24.3 s with 2, 23.9 s with 4)
    A.sh_on = 0;                       // set by launch_kind from the context (the environment is read once, at fig_ctx_create)
    return A;
}

// kind: 0 sequential fill, 1 begin, 2 eval (items), 3 end (list), 4 probe (reach bits)
template <bool LDS_TAB, int NT>
static hipError_t launch_kind(fig_ctx *ctx, const fig_ctx::Cls &c, const FigDevBatch &db, hipStream_t stream, int kind, int blocks, const void *list, int n, int qsel) {
    FigKernArgs A = kargs_of(c);
    A.qsel = qsel; A.sh_on = ctx->sh_on;
    hipError_t e = hipSuccess;
    if (kind == 0) {
        auto k = fig_fill_kernel<LDS_TAB, NT>;
        e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.c.lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(NT), c.c.lds, stream, ctx->dm, db, A);
    } else if (kind == 1) {
        auto k = fig_begin_kernel<LDS_TAB, NT>;
        e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.c.lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(NT), c.c.lds, stream, ctx->dm, db, A);
    } else if (kind == 2) {
        auto k = fig_eval_kernel<LDS_TAB, NT>;
        e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.c.lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(NT), c.c.lds, stream, ctx->dm, db, A, (const int4 *)list, n);
    } else if (kind == 4) {
        auto k = fig_probe_kernel<LDS_TAB, NT>;
        e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.c.lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(NT), c.c.lds, stream, ctx->dm, db, A);
    } else {
        auto k = fig_end_kernel<LDS_TAB, NT>;
        e = hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c.c.lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(NT), c.c.lds, stream, ctx->dm, db, A, (const int *)list, n);
    }
    return hipGetLastError();
}

static hipError_t launch_any(fig_ctx *ctx, const fig_ctx::Cls &c, const FigDevBatch &db, hipStream_t stream, int kind, int blocks, const void *list, int n, int qsel) {
    if (c.c.tiles > 0) return launch_kind<true, 512>(ctx, c, db, stream, kind, blocks, list, n, qsel);      // LDS-tiled: the LDS code path with a streamed table
    if (c.c.lds_tab) return c.c.nt == 256 ? launch_kind<true, 256>(ctx, c, db, stream, kind, blocks, list, n, qsel) : launch_kind<true, 512>(ctx, c, db, stream, kind, blocks, list, n, qsel);
    return launch_kind<false, 512>(ctx, c, db, stream, kind, blocks, list, n, qsel);
}

// Candidate-parallel scheduling of one class (see fig_engine_sched.h).  Host-driven rounds: begin -> {eval chunk,
// replay}* -> end.  Returns the number of kernel launches, or -1 on a HIP error (ctx->last_hip set).
static int run_class_parallel(fig_ctx *ctx, const fig_ctx::Cls &c, FigLane &ln) {
    FigDevBatch db = ctx->db;
    db.queue_head = ln.queue_head; db.scratch = ln.scratch;
    hipStream_t stream = ln.stream;
    hipSetDevice(ctx->device);
    const int n_cls = c.c.q_end - c.c.q_begin;
    int nl = 0;
    hipError_t e;
    auto fail = [&](hipError_t er) { ctx->last_hip = (int)er; return -1; };
    // every persistent launch pops from one of the lane's two queue heads and zeroes the other for its successor
    if ((e = launch_any(ctx, c, db, stream, 1, std::min(std::max(1, c.capacity), n_cls), nullptr, 0, ln.qsel)) != hipSuccess) return fail(e);
    ln.qsel ^= 1;
    nl++;
    int32_t *ctl = ln.h_ctl;
    const size_t ctl_n = (size_t)ctx->n_gaps * 4;
    std::vector<int> ids(ctx->h_order.begin() + c.c.q_begin, ctx->h_order.begin() + c.c.q_end);   // cost-sorted
    std::vector<int> items, entries, endlist;
    const int slots_cap = ctx->nslots;
    const int capacity = std::max(1, c.capacity);      // workgroups the device holds for this class (not capped by the gap count)
    const bool log = getenv("FIG_SCHED_LOG") != nullptr;
    const int minc = getenv("FIG_MIN_CHUNK") ? std::max(1, atoi(getenv("FIG_MIN_CHUNK"))) : 16;      // candidates per gap and round, at least
    const double ipw_base = getenv("FIG_ITEMS_PER_WG") ? std::max(1.0, atof(getenv("FIG_ITEMS_PER_WG"))) : (ctx->dm.unmapped ? 12.0 : 6.0);  // items per resident workgroup and round (partial-mode pass, 8 192 gaps: 6 -> 4 052 gaps/s, 12 -> 3 887, 24 -> 3 875, 48 -> 3 668: shorter rounds discard fewer candidates past an early stop; unmapped, measured on the bench batch: 4 -> 34.4 s, 8 -> 29.7, 12 -> 29.1, 16 -> 29.2 per step)
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_start = now();
    double t_prev = now(); int round = 0, last_items = 0, last_active = 0, last_chunk = 0, n_active_max = 0;
    while (true) {
        if ((e = hipMemcpyAsync(ctl, db.gapctl, ctl_n * 4, hipMemcpyDeviceToHost, stream)) != hipSuccess) return fail(e);
        if ((e = hipStreamSynchronize(stream)) != hipSuccess) return fail(e);
        int n_active = 0;
        for (int g : ids) if (ctl[(size_t)g * 4] == 1) n_active++;
        if (log) { double t = now(); fprintf(stderr, "[figsched] capG=%d round %d: active=%d chunk=%d items=%d blocks=%d  %.1f ms\n", c.c.capG, round, last_active, last_chunk, last_items, capacity, t - t_prev); t_prev = t; }
        round++;
        if (n_active == 0) break;
        // Candidates per gap this round: proportional to the candidates the gap still has, so that all gaps of the class
        // finish in about the same round and every round carries ~12 items per resident workgroup (of both lanes of a split class).  A gap that stops
        // early discards at most chunk-1 evaluations.
        long long rem_total = 0;
        for (int g : ids) if (ctl[(size_t)g * 4] == 1) rem_total += std::max(0, ctl[(size_t)g * 4 + 2] - ctl[(size_t)g * 4 + 1]);
        // Round size: `ipw_base` items per resident workgroup while the lane has about a hundred active gaps (the 512-gap bench
        // batch: 12 is its optimum), growing with the lane's number of active gaps (its maximum so far) up to 8x.  Measured on one box (round 3,
        // profiles/round3/largefill_*): the 2048-gap fill of the bench recipe takes 111.9 s with 12 items per workgroup and
        // round (121 rounds per lane of 1920 items over 385 active gaps: every round ends with a tail of the long items of
        // the most expensive gaps), 99.8 s with 48 and 98.5 s with 96 -- while 48 costs the 512-gap batch 3 %.
        n_active_max = std::max(n_active_max, n_active);
        const double ipw = ipw_base * std::min(8.0, std::max(1.0, n_active_max / 96.0));      // by the lane's largest active set: the rounds stay long to the end of a big fill
        const double share = rem_total > 0 ? (ipw * capacity / (double)std::max(1, c.c.nsplit)) / (double)rem_total : 1.0;
        // Admission: a gap gets at least `minc` candidates in a round it takes part in, and gaps are admitted in cost order
        // until the round is full.  With thousands of active gaps the proportional share alone would hand every gap a few
        // candidates per round: hundreds of rounds per gap and, worse, hundreds of DIFFERENT gaps in flight at once, whose
        // reads then miss the L2 (a 2048-gap fill ran at 0.33 of peak against 0.37 for 512 gaps).  Workgroups that pop
        // neighbouring items work on the same gap's reads.
        const double target = ipw * capacity / (double)std::max(1, c.c.nsplit);
        long long total = 0;
        int chunk = 0;
        items.clear(); entries.clear();
        for (int g : ids) {
            if (ctl[(size_t)g * 4] != 1) continue;
            if ((double)total >= 1.25 * target) break;           // the rest waits for a later round
            int j = ctl[(size_t)g * 4 + 1], range = ctl[(size_t)g * 4 + 2];
            int want = (int)std::ceil((range - j) * share);
            want = std::max(minc, std::min(want, slots_cap));
            want = std::min(want, slots_cap);
            int n = std::min(want, range - j);
            if (n <= 0) { n = 0; }                                // (replayed with nothing to evaluate: the replay closes the gap)
            chunk = std::max(chunk, n);
            total += n;
            entries.push_back(g); entries.push_back(n); entries.push_back(0); entries.push_back(0);
        }
        // items gap-major in descending-cost gap order (longest processing time first keeps the round's tail short)
        for (size_t q = 0; q < entries.size(); q += 4)
            for (int k = entries[q + 1] - 1; k >= 0; k--) { int g = entries[q]; items.push_back(g); items.push_back(ctl[(size_t)g * 4 + 1] + k); items.push_back(k); items.push_back(0); }
        if (ln.cap < items.size() || ln.cap < entries.size()) return fail(hipErrorOutOfMemory);
        int n_items = (int)(items.size() / 4), n_ent = (int)(entries.size() / 4);
        last_items = n_items; last_active = n_active; last_chunk = chunk;
        if (n_items > 0) {
            memcpy(ln.h_items, items.data(), items.size() * 4);
            if ((e = hipMemcpyAsync(ln.d_items, ln.h_items, items.size() * 4, hipMemcpyHostToDevice, stream)) != hipSuccess) return fail(e);
            if ((e = launch_any(ctx, c, db, stream, 2, std::min(capacity, n_items), ln.d_items, n_items, ln.qsel)) != hipSuccess) return fail(e);
            ln.qsel ^= 1;
            nl++;
        }
        memcpy(ln.h_entries, entries.data(), entries.size() * 4);
        if ((e = hipMemcpyAsync(ln.d_entries, ln.h_entries, entries.size() * 4, hipMemcpyHostToDevice, stream)) != hipSuccess) return fail(e);
        hipLaunchKernelGGL(fig_replay_kernel, dim3(n_ent), dim3(64), sizeof(FigState) + 64, stream, ctx->dm, db, (const int4 *)ln.d_entries, n_ent);
        if ((e = hipGetLastError()) != hipSuccess) return fail(e);
        nl++;
    }
    endlist.clear();
    for (int g : ids) if (ctl[(size_t)g * 4] == 2) endlist.push_back(g);
    if (!endlist.empty()) {
        if ((e = hipStreamSynchronize(stream)) != hipSuccess) return fail(e);      // h_items is reused: the last round's upload must have left it
        memcpy(ln.h_items, endlist.data(), endlist.size() * 4);
        if ((e = hipMemcpyAsync(ln.d_items, ln.h_items, endlist.size() * 4, hipMemcpyHostToDevice, stream)) != hipSuccess) return fail(e);
        if ((e = launch_any(ctx, c, db, stream, 3, std::min(capacity, (int)endlist.size()), ln.d_items, (int)endlist.size(), ln.qsel)) != hipSuccess) return fail(e);
        ln.qsel ^= 1;
        nl++;
        if (log) { const double t0 = now(); hipStreamSynchronize(stream); fprintf(stderr, "[figsched] capG=%d end kernel: %d gaps, %.1f ms (lane done at %.1f ms since its first round)\n", c.c.capG, (int)endlist.size(), now() - t0, now() - t_start); }
    }
    (void)n_cls;
    return nl;
}

extern "C" int fig_fill_resident(fig_ctx *ctx, fig_gap_results *out) {
    if (!ctx || !out || !ctx->have_batch) return FIG_EINVAL;
    if (!out->filled_len || !out->gaptofill || !out->str_off || (!out->str && ctx->str_total > 0)) return FIG_EINVAL;
    hipSetDevice(ctx->device);
    int64_t ng = ctx->n_gaps;
    FigDevBatch &db = ctx->db;
    // optional debug / draw planes live in per-call device buffers
    std::vector<void *> tmp;
    auto talloc = [&](size_t n, void **p) -> int { if (hipMalloc(p, n ? n : 8) != hipSuccess) return FIG_ENOMEM; tmp.push_back(*p); return FIG_OK; };
    auto tfree = [&]() { for (void *p : tmp) hipFree(p); };
    db.dbg_n_cand = nullptr; db.dbg_cand_i = nullptr; db.dbg_cand_lik = nullptr; db.dbg_max_cand = 0; db.dbg_n_place = nullptr;
    db.draw_pos = db.draw_isz = db.draw_len = nullptr;
    db.dbg_counts = db.dbg_read_maxlv = nullptr; db.dbg_plane_cols = db.dbg_plane_reads = 0;
    void *p;
    if (out->dbg_n_cand && out->dbg_cand_i && out->dbg_cand_lik && out->dbg_max_cand > 0) {
        db.dbg_max_cand = out->dbg_max_cand;
        if (talloc((size_t)ng * 4, &p)) { tfree(); return FIG_ENOMEM; } db.dbg_n_cand = (int32_t *)p;
        if (talloc((size_t)ng * out->dbg_max_cand * 12, &p)) { tfree(); return FIG_ENOMEM; } db.dbg_cand_i = (int32_t *)p;
        if (talloc((size_t)ng * out->dbg_max_cand * 8, &p)) { tfree(); return FIG_ENOMEM; } db.dbg_cand_lik = (double *)p;
        hipMemsetAsync(db.dbg_n_cand, 0, (size_t)ng * 4, ctx->stream);
        if (out->dbg_n_place) { if (talloc((size_t)ng * 4, &p)) { tfree(); return FIG_ENOMEM; } db.dbg_n_place = (int32_t *)p; hipMemsetAsync(db.dbg_n_place, 0, (size_t)ng * 4, ctx->stream); }
        if (out->dbg_counts && out->dbg_plane_cols > 0) {
            const size_t n = (size_t)ng * out->dbg_max_cand * out->dbg_plane_cols * 5 * 8;
            if (talloc(n, &p)) { tfree(); return FIG_ENOMEM; } db.dbg_counts = (double *)p; db.dbg_plane_cols = out->dbg_plane_cols;
            hipMemsetAsync(db.dbg_counts, 0, n, ctx->stream);
        }
        if (out->dbg_read_maxlv && out->dbg_plane_reads > 0) {
            const size_t n = (size_t)ng * out->dbg_max_cand * out->dbg_plane_reads * 8;
            if (talloc(n, &p)) { tfree(); return FIG_ENOMEM; } db.dbg_read_maxlv = (double *)p; db.dbg_plane_reads = out->dbg_plane_reads;
            hipMemsetAsync(db.dbg_read_maxlv, 0, n, ctx->stream);
        }
    }
    int64_t nr = ctx->n_ureads + ctx->n_preads;
    if (out->draw_pos && out->draw_isz && out->draw_len) {
        if (talloc((size_t)nr * 4, &p)) { tfree(); return FIG_ENOMEM; } db.draw_pos = (int32_t *)p;
        if (talloc((size_t)nr * 4, &p)) { tfree(); return FIG_ENOMEM; } db.draw_isz = (int32_t *)p;
        if (talloc((size_t)ng * 8, &p)) { tfree(); return FIG_ENOMEM; } db.draw_len = (int32_t *)p;
        hipMemsetAsync(db.draw_isz, 0, (size_t)nr * 4, ctx->stream);
    }
    hipMemsetAsync(db.counters, 0, 512, ctx->stream);
    // a previous call that failed half-way may have left queue heads / ping-pong selectors inconsistent: start clean
    hipMemsetAsync(db.queue_head, 0, 64 * (ctx->classes.size() + 1), ctx->stream);
    for (auto &l : ctx->lanes) l.qsel = 0;
    hipMemsetAsync(db.filled_len, 0, (size_t)std::max<int64_t>(ng, 1) * 4, ctx->stream);
    hipMemsetAsync(db.gaptofill, 0, (size_t)std::max<int64_t>(ng, 1) * 4, ctx->stream);
    hipMemsetAsync(db.str, 'N', (size_t)ctx->str_total, ctx->stream);
    hipMemsetAsync(db.gapctl, 0, (size_t)std::max<int64_t>(ng, 1) * 16, ctx->stream);
    hipEventRecord(ctx->ev0, ctx->stream);
    int nl = 0;
    const char *sched = getenv("FIG_SCHED");
    const bool seq = sched && strcmp(sched, "seq") == 0;
    if (seq) {
        for (size_t ci = 0; ci < ctx->classes.size(); ci++) {
            const fig_ctx::Cls &c = ctx->classes[ci];
            hipMemsetAsync(db.queue_head, 0, 4, ctx->stream);
            hipError_t e = launch_any(ctx, c, db, ctx->stream, 0, c.blocks, nullptr, 0);
            if (e != hipSuccess) { ctx->last_hip = (int)e; tfree(); return FIG_EHIP; }
            nl++;
        }
    } else {
        // one host thread + stream per class; every lane starts after ev0 and the main stream joins them before ev1
        const size_t nc = ctx->classes.size();
        std::vector<int> rcs(nc, 0);
        ctx->db = db;
        for (size_t ci = 0; ci < nc; ci++) hipStreamWaitEvent(ctx->lanes[ci].stream, ctx->ev0, 0);
        const char *ser = getenv("FIG_LANES");
        if (nc <= 1 || (ser && strcmp(ser, "serial") == 0)) {
            for (size_t ci = 0; ci < nc; ci++) rcs[ci] = run_class_parallel(ctx, ctx->classes[ci], ctx->lanes[ci]);
        } else {
            std::vector<std::thread> th;
            for (size_t ci = 0; ci < nc; ci++) th.emplace_back([&, ci] { rcs[ci] = run_class_parallel(ctx, ctx->classes[ci], ctx->lanes[ci]); });
            for (auto &t : th) t.join();
        }
        for (size_t ci = 0; ci < nc; ci++) {
            if (rcs[ci] < 0) { hipDeviceSynchronize(); tfree(); return FIG_EHIP; }
            nl += rcs[ci];
            hipEventRecord(ctx->lanes[ci].done, ctx->lanes[ci].stream);
            hipStreamWaitEvent(ctx->stream, ctx->lanes[ci].done, 0);
        }
    }
    hipEventRecord(ctx->ev1, ctx->stream);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { ctx->last_hip = (int)e; tfree(); return FIG_EHIP; }
    float ms = 0; hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    ctx->stats.kernel_ms = ms; ctx->stats.n_launches = nl;
    // ---- results back
    hipEventRecord(ctx->ev0, ctx->stream);
    std::vector<char> hstr((size_t)ctx->str_total);
    hipMemcpyAsync(out->filled_len, db.filled_len, (size_t)ng * 4, hipMemcpyDeviceToHost, ctx->stream);
    hipMemcpyAsync(out->gaptofill, db.gaptofill, (size_t)ng * 4, hipMemcpyDeviceToHost, ctx->stream);
    if (ctx->str_total) hipMemcpyAsync(hstr.data(), db.str, (size_t)ctx->str_total, hipMemcpyDeviceToHost, ctx->stream);
    unsigned long long cnt[64] = {0};
    hipMemcpyAsync(cnt, db.counters, 512, hipMemcpyDeviceToHost, ctx->stream);
    if (db.dbg_n_cand) {
        hipMemcpyAsync(out->dbg_n_cand, db.dbg_n_cand, (size_t)ng * 4, hipMemcpyDeviceToHost, ctx->stream);
        hipMemcpyAsync(out->dbg_cand_i, db.dbg_cand_i, (size_t)ng * out->dbg_max_cand * 12, hipMemcpyDeviceToHost, ctx->stream);
        hipMemcpyAsync(out->dbg_cand_lik, db.dbg_cand_lik, (size_t)ng * out->dbg_max_cand * 8, hipMemcpyDeviceToHost, ctx->stream);
        if (db.dbg_n_place) hipMemcpyAsync(out->dbg_n_place, db.dbg_n_place, (size_t)ng * 4, hipMemcpyDeviceToHost, ctx->stream);
        if (db.dbg_counts) hipMemcpyAsync(out->dbg_counts, db.dbg_counts, (size_t)ng * out->dbg_max_cand * out->dbg_plane_cols * 5 * 8, hipMemcpyDeviceToHost, ctx->stream);
        if (db.dbg_read_maxlv) hipMemcpyAsync(out->dbg_read_maxlv, db.dbg_read_maxlv, (size_t)ng * out->dbg_max_cand * out->dbg_plane_reads * 8, hipMemcpyDeviceToHost, ctx->stream);
    }
    if (db.draw_pos) {
        hipMemcpyAsync(out->draw_pos, db.draw_pos, (size_t)nr * 4, hipMemcpyDeviceToHost, ctx->stream);
        hipMemcpyAsync(out->draw_isz, db.draw_isz, (size_t)nr * 4, hipMemcpyDeviceToHost, ctx->stream);
        hipMemcpyAsync(out->draw_len, db.draw_len, (size_t)ng * 8, hipMemcpyDeviceToHost, ctx->stream);
    }
    hipEventRecord(ctx->ev1, ctx->stream);
    e = hipStreamSynchronize(ctx->stream);
    tfree();
    if (e != hipSuccess) { ctx->last_hip = (int)e; return FIG_EHIP; }
    hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1);
    ctx->stats.d2h_ms = ms;
#ifdef FIG_PROF
    { const char *nm[14] = {"A.wait", "B.wait", "-", "M.finish", "PR.pre", "PR.estep", "PR.mid", "PR.mle", "PR.post", "A.chain", "A.logexp", "A.work", "B.work", "M.chains"};
      for (int i = 0; i < 14; i++) fprintf(stderr, "[figprof] %-9s %8.1f Gcycles = %5.1f %% of wave-cycles\n", nm[i], cnt[8 + i] / 1e9, cnt[31] ? 100.0 * cnt[8 + i] / cnt[31] : 0.0);
      { const char *mn[4] = {"M.setup", "M.hint", "M.rounds", "M.surv"};
        for (int i = 0; i < 4; i++) fprintf(stderr, "[figprof] %-9s %8.1f Gcycles = %5.1f %% of wave-cycles\n", mn[i], cnt[22 + i] / 1e9, cnt[31] ? 100.0 * cnt[22 + i] / cnt[31] : 0.0); }
      fprintf(stderr, "[figprof] raw slots:"); for (int i = 0; i < 22; i++) fprintf(stderr, " %d:%.1f", i, cnt[8 + i] / 1e9); for (int i = 22; i < 40; i++) fprintf(stderr, " %d:%.1f", i, cnt[32 + i - 22] / 1e9); fprintf(stderr, "\n");
      fprintf(stderr, "[figprof] barrier wait %.3f of %.3f wave-Gcycles = %.1f %%\n", cnt[30] / 1e9, cnt[31] / 1e9, cnt[31] ? 100.0 * cnt[30] / cnt[31] : 0.0); }
#endif
    if (getenv("FIG_SCHED_LOG")) fprintf(stderr, "[figsched] useful flops %.4g, speculative evaluations executed %.4g (%.1f %% discarded)\n", (double)cnt[1], (double)cnt[2], cnt[2] ? 100.0 * (1.0 - ((double)cnt[1] / (double)cnt[2])) : 0.0);
    ctx->stats.place_calls = (int64_t)cnt[0];
    ctx->stats.alg_flops = (double)cnt[1];
    ctx->stats.spec_flops = (double)cnt[2];
    ctx->stats.mle_alg_flops = (double)cnt[3];
    ctx->stats.mle_exec_flops = (double)cnt[4];
    // compact strings
    int64_t need = 0;
    for (int64_t g = 0; g < ng; g++) need += out->filled_len[g] > 0 ? out->filled_len[g] : 0;
    if (need > out->str_capacity) return FIG_ENOSPC;
    int64_t o = 0;
    for (int64_t g = 0; g < ng; g++) {
        out->str_off[g] = o;
        int n = out->filled_len[g];
        if (n > 0) { memcpy(out->str + o, hstr.data() + ctx->h_str_off[g], (size_t)n); o += n; }
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

extern "C" int fig_get_stats(const fig_ctx *ctx, fig_stats *out) {
    if (!ctx || !out) return FIG_EINVAL;
    *out = ctx->stats;
    return FIG_OK;
}
