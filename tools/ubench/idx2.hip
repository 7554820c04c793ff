// Micro-benchmark 2 (clean loop bodies): wave-uniform 8-way operand select through VGPR index mode vs plain multiplies.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define C 24
struct alignas(16) PQ { double p, q; };
#define M0MUL(pa, pb, w) \
    "s_and_b32 m0, " w ", 0xffff\n v_mul_f64 " pa ", v[232:233], " pa "\n" \
    "s_lshr_b32 m0, " w ", 16\n v_mul_f64 " pb ", v[232:233], " pb "\n"
#define NOMUL(pa, pb, w) \
    "v_mul_f64 " pa ", v[232:233], " pa "\n v_mul_f64 " pb ", v[234:235], " pb "\n"
#define P(i) "%[p" #i "]"
#define W(i) "%[w" #i "]"
#define BODY(M) M(P(0), P(1), W(0)) M(P(2), P(3), W(1)) M(P(4), P(5), W(2)) M(P(6), P(7), W(3)) M(P(8), P(9), W(4)) M(P(10), P(11), W(5)) \
                M(P(12), P(13), W(6)) M(P(14), P(15), W(7)) M(P(16), P(17), W(8)) M(P(18), P(19), W(9)) M(P(20), P(21), W(10)) M(P(22), P(23), W(11))
#define PO(i) [p##i] "+v"(p[i])
#define WI(i) [w##i] "s"(w[i])
#define OPS : PO(0), PO(1), PO(2), PO(3), PO(4), PO(5), PO(6), PO(7), PO(8), PO(9), PO(10), PO(11), PO(12), PO(13), PO(14), PO(15), PO(16), PO(17), PO(18), PO(19), PO(20), PO(21), PO(22), PO(23) \
    : "{v[232:233]}"(f0), "{v[234:235]}"(f1), "{v[236:237]}"(f2), "{v[238:239]}"(f3), "{v[240:241]}"(f4), "{v[242:243]}"(f5), "{v[244:245]}"(f6), "{v[246:247]}"(f7), \
      WI(0), WI(1), WI(2), WI(3), WI(4), WI(5), WI(6), WI(7), WI(8), WI(9), WI(10), WI(11) : "m0", "scc"

template <int V>
__global__ void __launch_bounds__(512) k(const PQ *tab, int ncol, const uint32_t *idx, int nsteps, double *out) {
    extern __shared__ double lds[];
    PQ *T = (PQ *)lds;
    for (int i = threadIdx.x; i < 4 * ncol; i += blockDim.x) T[i] = tab[i];
    __syncthreads();
    const int o = threadIdx.x;
    double p[C];
#pragma unroll
    for (int s = 0; s < C; s++) p[s] = 1.0 + 1e-3 * s;
    uint32_t w[C / 2];
#pragma unroll
    for (int k2 = 0; k2 < C / 2; k2++) w[k2] = __builtin_amdgcn_readfirstlane(idx[k2]);
    const double omf = 0.996, ef = 0.004, omr = 0.995, er = 0.005;
    PQ an[4];
    { const PQ *col = T + o; an[0] = col[0]; an[1] = col[ncol]; an[2] = col[2 * ncol]; an[3] = col[3 * ncol]; }
    for (int j = 0; j < nsteps; j++) {
        const PQ a0 = an[0], a1 = an[1], a2 = an[2], a3 = an[3];
        { const PQ *col = T + (o + ((j + 1) & 127)); an[0] = col[0]; an[1] = col[ncol]; an[2] = col[2 * ncol]; an[3] = col[3 * ncol]; }
        const double f0 = a0.p * omf + ef * a0.q, f1 = a1.p * omf + ef * a1.q, f2 = a2.p * omf + ef * a2.q, f3 = a3.p * omf + ef * a3.q;
        const double f4 = a0.p * omr + er * a0.q, f5 = a1.p * omr + er * a1.q, f6 = a2.p * omr + er * a2.q, f7 = a3.p * omr + er * a3.q;
        if (V == 0) {
#pragma unroll
            for (int s = 0; s < C; s++) p[s] = ((s & 1) ? f1 : f6) * p[s];
            asm volatile("" :: "v"(f0), "v"(f2), "v"(f3), "v"(f4), "v"(f5), "v"(f7));
        } else if (V == 1) {
            asm volatile("s_set_gpr_idx_on %[w0], 1\n" BODY(M0MUL) "s_set_gpr_idx_off\n" OPS);
        } else if (V == 2) {
            asm volatile(BODY(NOMUL) OPS);
        } else {   // select through v_cndmask (what plain C gives)
            const double F[8] = {f0, f1, f2, f3, f4, f5, f6, f7};
#pragma unroll
            for (int s = 0; s < C; s++) {
                const uint32_t e = (w[s >> 1] >> ((s & 1) * 16)) & 0xfff;
                double f = F[0];
#pragma unroll
                for (int k2 = 1; k2 < 8; k2++) f = (e == 2u * k2) ? F[k2] : f;
                p[s] = f * p[s];
            }
        }
    }
#pragma unroll
    for (int s = 0; s < C; s++) out[((long long)blockIdx.x * blockDim.x + threadIdx.x) * C + s] = p[s];
}

__global__ void kclk(long long *o, int n) {
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0000001;
    long long c0 = __builtin_readcyclecounter(); unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; i++) { a = a * b; a = a * b; a = a * b; a = a * b; a = a * b; a = a * b; a = a * b; a = a * b; }
    long long c1 = __builtin_readcyclecounter(); unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { o[blockIdx.x * 2] = c1 - c0; o[blockIdx.x * 2 + 1] = (long long)(r1 - r0); }
    if (a == 0.5) o[0] = 0;
}
template <int V> void run(const PQ *dt, int ncol, const uint32_t *di, int nsteps, double *dout, int threads, int nblk, std::vector<double> *keep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<V>, dim3(nblk), dim3(threads), 4 * ncol * sizeof(PQ), 0, dt, ncol, di, nsteps, dout);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    if (hipGetLastError() != hipSuccess) printf("V%d: launch error\n", V);
    const double mults = (double)nblk * threads * C * nsteps;
    const double cyc_per_step_wave = ms * 1e-3 * 2.4e9 / nsteps;
    printf("V%d threads %d: %.3f ms, %.2f Tmul/s, %.0f cycles@2.4GHz per step (FP64 ops/step/lane: %d)\n", V, threads, ms, mults / ms * 1e-9, cyc_per_step_wave, V == 3 ? 48 : 48);
    if (keep) { keep->resize((size_t)nblk * threads * C); hipMemcpy(keep->data(), dout, keep->size() * 8, hipMemcpyDeviceToHost); }
}

int main() {
    const int ncol = 1100, nsteps = 6000, nblk = 256;
    std::vector<PQ> tab(4 * ncol);
    srand(1);
    for (auto &v : tab) { v.p = 1.0 - 1e-9 * (rand() & 1023); v.q = 0.01 * (rand() / (double)RAND_MAX); }
    std::vector<uint32_t> idx(C / 2);
    for (auto &w : idx) { uint32_t a = 0x1000u | (2u * (rand() & 7)), b = 0x1000u | (2u * (rand() & 7)); w = a | (b << 16); }
    PQ *dt; uint32_t *di; double *dout;
    hipMalloc(&dt, tab.size() * sizeof(PQ)); hipMalloc(&di, idx.size() * 4);
    hipMemcpy(dt, tab.data(), tab.size() * sizeof(PQ), hipMemcpyHostToDevice); hipMemcpy(di, idx.data(), idx.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&dout, (size_t)nblk * 512 * C * 8);
    { long long *dc; hipMalloc(&dc, 256 * 16); hipLaunchKernelGGL(kclk, dim3(256), dim3(512), 0, 0, dc, 200000); hipDeviceSynchronize();
      long long h[4]; hipMemcpy(h, dc, 32, hipMemcpyDeviceToHost);
      printf("clock probe: s_memtime ticks %lld, s_memrealtime ticks (100 MHz) %lld -> s_memtime at %.1f MHz; 1.6e6 dependent v_mul_f64 per lane: %.2f memtime ticks each, %.2f ns each\n", h[0], h[1], 100.0 * h[0] / h[1], h[0] / 1.6e6, h[1] * 10.0 / 1.6e6); }
    for (int threads : {256, 512}) {
        std::vector<double> r1, r3;
        run<0>(dt, ncol, di, nsteps, dout, threads, nblk, nullptr);
        run<1>(dt, ncol, di, nsteps, dout, threads, nblk, &r1);
        run<2>(dt, ncol, di, nsteps, dout, threads, nblk, nullptr);
        run<3>(dt, ncol, di, nsteps, dout, threads, nblk, &r3);
        size_t bad = 0; for (size_t i = 0; i < r1.size(); i++) if (r1[i] != r3[i]) bad++;
        printf("threads %d: index-mode vs cndmask select mismatches: %zu of %zu\n", threads, bad, r1.size());
    }
    return 0;
}
