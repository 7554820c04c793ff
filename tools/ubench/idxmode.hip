// Micro-benchmark: VGPR index mode (s_set_gpr_idx / M0) as a wave-uniform 8-way operand select for FP64 multiplies.
// One lane = one placement; C products (one per read of a chunk) per lane; per chain step the lane computes the eight
// factors f[orientation][base] of its column once (24 FP64 ops) and every read multiplies its product by the factor its
// own base selects (1 SALU + 1 VALU).  Checks the result against a plain C select and reports cycles per step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <cmath>

#define C 24
struct alignas(16) PQ { double p, q; };

// products p0..p23 *= F[idx_s]; F lives in v[232:247] (8 doubles); idx stream: 16-bit entries 0x1000 | (2 * k)
#define MUL2(pa, pb, w) \
    "s_and_b32 m0, " w ", 0xffff\n v_mul_f64 " pa ", v[232:233], " pa "\n" \
    "s_lshr_b32 m0, " w ", 16\n v_mul_f64 " pb ", v[232:233], " pb "\n"

#define MULI2(pa, pb, w) \
    "s_set_gpr_idx_idx " w "\n v_mul_f64 " pa ", v[232:233], " pa "\n" \
    "s_set_gpr_idx_idx " w "\n v_mul_f64 " pb ", v[232:233], " pb "\n"
#define MULN2(pa, pb, w) \
    "v_mul_f64 " pa ", v[232:233], " pa "\n" \
    "v_mul_f64 " pb ", v[232:233], " pb "\n"
#define MULP2(pa, pb, w) \
    "s_and_b32 m0, " w ", 0xffff\n v_mul_f64 " pa ", v[232:233], " pa "\n" \
    "v_mul_f64 " pb ", v[232:233], " pb "\n"
#define DEFMUL(NAME, M) \
__device__ __forceinline__ void NAME(double &p0, double &p1, double &p2, double &p3, double &p4, double &p5, double &p6, double &p7, \
                                      double &p8, double &p9, double &p10, double &p11, \
                                      double f0, double f1, double f2, double f3, double f4, double f5, double f6, double f7, \
                                      uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t w4, uint32_t w5) { \
    asm volatile( \
        "s_set_gpr_idx_on %[w0], 1\n" \
        M("%[p0]", "%[p1]", "%[w0]") M("%[p2]", "%[p3]", "%[w1]") M("%[p4]", "%[p5]", "%[w2]") \
        M("%[p6]", "%[p7]", "%[w3]") M("%[p8]", "%[p9]", "%[w4]") M("%[p10]", "%[p11]", "%[w5]") \
        "s_set_gpr_idx_off\n" \
        : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [p4] "+v"(p4), [p5] "+v"(p5), [p6] "+v"(p6), [p7] "+v"(p7), \
          [p8] "+v"(p8), [p9] "+v"(p9), [p10] "+v"(p10), [p11] "+v"(p11) \
        : "{v[232:233]}"(f0), "{v[234:235]}"(f1), "{v[236:237]}"(f2), "{v[238:239]}"(f3), "{v[240:241]}"(f4), "{v[242:243]}"(f5), \
          "{v[244:245]}"(f6), "{v[246:247]}"(f7), \
          [w0] "s"(w0), [w1] "s"(w1), [w2] "s"(w2), [w3] "s"(w3), [w4] "s"(w4), [w5] "s"(w5) \
        : "m0"); }
DEFMUL(mul12i, MULI2)
DEFMUL(mul12n, MULN2)
DEFMUL(mul12p, MULP2)
__device__ __forceinline__ void mul12(double &p0, double &p1, double &p2, double &p3, double &p4, double &p5, double &p6, double &p7,
                                      double &p8, double &p9, double &p10, double &p11,
                                      double f0, double f1, double f2, double f3, double f4, double f5, double f6, double f7,
                                      uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t w4, uint32_t w5) {
    asm volatile(
        "s_set_gpr_idx_on %[w0], 1\n"
        MUL2("%[p0]", "%[p1]", "%[w0]") MUL2("%[p2]", "%[p3]", "%[w1]") MUL2("%[p4]", "%[p5]", "%[w2]")
        MUL2("%[p6]", "%[p7]", "%[w3]") MUL2("%[p8]", "%[p9]", "%[w4]") MUL2("%[p10]", "%[p11]", "%[w5]")
        "s_set_gpr_idx_off\n"
        : [p0] "+v"(p0), [p1] "+v"(p1), [p2] "+v"(p2), [p3] "+v"(p3), [p4] "+v"(p4), [p5] "+v"(p5), [p6] "+v"(p6), [p7] "+v"(p7),
          [p8] "+v"(p8), [p9] "+v"(p9), [p10] "+v"(p10), [p11] "+v"(p11)
        : "{v[232:233]}"(f0), "{v[234:235]}"(f1), "{v[236:237]}"(f2), "{v[238:239]}"(f3), "{v[240:241]}"(f4), "{v[242:243]}"(f5),
          "{v[244:245]}"(f6), "{v[246:247]}"(f7),
          [w0] "s"(w0), [w1] "s"(w1), [w2] "s"(w2), [w3] "s"(w3), [w4] "s"(w4), [w5] "s"(w5)
        : "m0");
}

extern "C" __global__ void __launch_bounds__(512) k_idx(const PQ *tab, int ncol, const uint32_t *idx, const double *kt, int nsteps, double *out, long long *cyc, int use_asm, int sload) {
    extern __shared__ double lds[];
    PQ *T = (PQ *)lds;
    for (int i = threadIdx.x; i < 4 * ncol; i += blockDim.x) T[i] = tab[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int o = wave * 64 + lane;
    double p[C];
#pragma unroll
    for (int s = 0; s < C; s++) p[s] = 1.0 + 1e-3 * s;
    typedef const uint32_t __attribute__((address_space(4))) *cu32p;
    typedef const double __attribute__((address_space(4))) *cdp;
    cu32p ix = (cu32p)idx; cdp kc = (cdp)kt;
    long long t0 = __builtin_readcyclecounter();
    uint32_t w[C / 2], wn[C / 2];
    double kk[4], kn[4];
    PQ a[4], an[4];
#pragma unroll
    for (int k = 0; k < C / 2; k++) wn[k] = ix[k];
#pragma unroll
    for (int k = 0; k < 4; k++) kn[k] = kc[k];
    { const PQ *col = T + o; an[0] = col[0]; an[1] = col[ncol]; an[2] = col[2 * ncol]; an[3] = col[3 * ncol]; }
    for (int j = 0; j < nsteps; j++) {
#pragma unroll
        for (int k = 0; k < C / 2; k++) w[k] = wn[k];
#pragma unroll
        for (int k = 0; k < 4; k++) { kk[k] = kn[k]; a[k] = an[k]; }
        {   // prefetch step j + 1 (scalar data loop-invariant in this variant: pure issue behaviour)
            if (sload) {
                cu32p wp = ix + ((j + 1) % 150) * (C / 2);
#pragma unroll
                for (int k = 0; k < C / 2; k++) wn[k] = wp[k];
#pragma unroll
                for (int k = 0; k < 4; k++) kn[k] = kc[4 * ((j + 1) % 150) + k];
            }
            const PQ *col = T + (o + ((j + 1) & 127));
            an[0] = col[0]; an[1] = col[ncol]; an[2] = col[2 * ncol]; an[3] = col[3 * ncol];
        }
        const double omf = kk[0], ef = kk[1], omr = kk[2], er = kk[3];
        const PQ a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
        const double f0 = a0.p * omf + ef * a0.q, f1 = a1.p * omf + ef * a1.q, f2 = a2.p * omf + ef * a2.q, f3 = a3.p * omf + ef * a3.q;
        const double f4 = a0.p * omr + er * a0.q, f5 = a1.p * omr + er * a1.q, f6 = a2.p * omr + er * a2.q, f7 = a3.p * omr + er * a3.q;
#define CALL2(FN) do { FN(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9], p[10], p[11], f0, f1, f2, f3, f4, f5, f6, f7, w[0], w[1], w[2], w[3], w[4], w[5]); \
            FN(p[12], p[13], p[14], p[15], p[16], p[17], p[18], p[19], p[20], p[21], p[22], p[23], f0, f1, f2, f3, f4, f5, f6, f7, w[6], w[7], w[8], w[9], w[10], w[11]); } while (0)
        if (use_asm == 1) CALL2(mul12);
        else if (use_asm == 2) CALL2(mul12i);
        else if (use_asm == 3) CALL2(mul12n);
        else if (use_asm == 4) CALL2(mul12p);
        else if (use_asm == 5) {
#pragma unroll
            for (int s = 0; s < C; s++) p[s] = ((s & 1) ? f1 : f6) * p[s];
        } else {
            const double F[8] = {f0, f1, f2, f3, f4, f5, f6, f7};
#pragma unroll
            for (int s = 0; s < C; s++) {
                const uint32_t e = (w[s >> 1] >> ((s & 1) * 16)) & 0xfff;
                double f = F[0];
#pragma unroll
                for (int k = 1; k < 8; k++) f = (e == 2u * k) ? F[k] : f;
                p[s] = f * p[s];
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    double acc = 0;
#pragma unroll
    for (int s = 0; s < C; s++) out[((long long)blockIdx.x * blockDim.x + threadIdx.x) * C + s] = p[s];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    (void)acc;
}

int main(int argc, char **argv) {
    const int ncol = 1100, nsteps = 6000, nblk = 256;
    std::vector<PQ> tab(4 * ncol);
    srand(1);
    for (auto &v : tab) { v.p = 1.0 - 1e-9 * (rand() & 1023); v.q = 0.01 * (rand() / (double)RAND_MAX); }
    std::vector<double> kt(4 * nsteps + 4);
    for (int j = 0; j < nsteps; j++) { kt[4 * j + 1] = 1e-5 * j * 0; kt[4 * j] = 1 - kt[4 * j + 1]; kt[4 * j + 3] = 0; kt[4 * j + 2] = 1 - kt[4 * j + 3]; }
    std::vector<uint32_t> idx((nsteps + 1) * (C / 2));
    for (auto &w : idx) { uint32_t a = 0x1000u | (2u * (rand() & 7)), b = 0x1000u | (2u * (rand() & 7)); w = a | (b << 16); }
    PQ *dt; uint32_t *di; double *dk, *dout; long long *dc;
    hipMalloc(&dt, tab.size() * sizeof(PQ)); hipMalloc(&di, idx.size() * 4); hipMalloc(&dk, kt.size() * 8);
    hipMemcpy(dt, tab.data(), tab.size() * sizeof(PQ), hipMemcpyHostToDevice); hipMemcpy(di, idx.data(), idx.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dk, kt.data(), kt.size() * 8, hipMemcpyHostToDevice);
    std::vector<std::vector<double>> res(16);
    for (int threads : {256, 512}) {
        hipMalloc(&dout, (size_t)nblk * threads * C * 8); hipMalloc(&dc, nblk * 8);
        for (int use_asm : {1, 3, 4, 5, 11, 15}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k_idx, dim3(nblk), dim3(threads), 4 * ncol * sizeof(PQ), 0, dt, ncol, di, dk, nsteps, dout, dc, use_asm % 10, use_asm / 10);
                hipEventRecord(e1); hipEventSynchronize(e1);
                if (hipGetLastError() != hipSuccess) { printf("launch error\n"); return 1; }
                float ms; hipEventElapsedTime(&ms, e0, e1);
                std::vector<long long> cyc(nblk); hipMemcpy(cyc.data(), dc, nblk * 8, hipMemcpyDeviceToHost);
                double mc = 0; for (auto c : cyc) mc += c; mc /= nblk;
                const double mults = (double)nblk * threads * C * nsteps;
                if (rep == 2) printf("threads %d asm %d: %.3f ms, %.2f Tmul/s, readcyclecounter/step %.1f (per product %.2f)\n", threads, use_asm, ms, mults / ms * 1e-9, mc / nsteps, mc / nsteps / C);
            }
            std::vector<double> r((size_t)nblk * threads * C); hipMemcpy(r.data(), dout, r.size() * 8, hipMemcpyDeviceToHost);
            res[use_asm] = r;
        }
        size_t bad = 0; for (size_t i = 0; i < res[0].size(); i++) if (res[3][i] != res[5][i]) bad++;
        printf("threads %d: asm vs C select mismatches: %zu of %zu (sample %.17g %.17g)\n", threads, bad, res[3].size(), res[1][5], res[11][5]);
        hipFree(dout); hipFree(dc);
    }
    return 0;
}
