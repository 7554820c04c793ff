// Micro-benchmark: the column pass of the E-step in isolation.  A 256-thread workgroup (4 waves = 4 bases), 80 KB of LDS so
// that two workgroups share a CU as in the engine; every wave adds, for each of `rows` weight rows, the n positions of its
// base: acc[i] += wrow[x + 64 i - j] (ds_read_b64 + v_add_f64 per column tile).  Reports wave-cycles per load+add for
//   V0: the engine's loop (4 positions: 4 CPL loads, then 4 CPL adds)
//   V1: the same with two register sets (loads of group k + 1 issued before the adds of group k)
//   V2: 8 positions per group (CPL <= 2)
//   V3: loads only (no adds): what the LDS delivers to this access pattern at this occupancy
//   V4: two register sets of 8 loads with hand-placed waits (inline-asm ds_read_b64, s_waitcnt lgkmcnt(8)): the compiler's own
//       waits in V1/V2 are lgkmcnt(0) at the loop's join points, which undoes the overlap
// and for 1 and 2 workgroups per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
extern __shared__ __attribute__((aligned(16))) double lds[];
typedef const volatile __attribute__((address_space(3))) double *ldsvp;
__device__ static __forceinline__ double ldv(const double *p) { return *(ldsvp)p; }

template <int CPL, int V>
__global__ void __launch_bounds__(256) k(double *out, unsigned long long *cyc, int rows, int n, int rowlen) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * rowlen; i += 256) lds[i] = 1.0 + 1e-9 * i;
    __syncthreads();
    // position list of this wave's base: n descending bytes, 4 per dword, one dword per lane
    uint32_t plv = 0;
    if (lane * 4 < n + 3) {
        for (int u = 0; u < 4; u++) { const int q = lane * 4 + u; const int j = q < n ? (149 - 4 * q - (wave & 3)) : 0; plv |= (uint32_t)(j < 0 ? 0 : j) << (8 * u); }
    }
    double acc[CPL];
#pragma unroll
    for (int i = 0; i < CPL; i++) acc[i] = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < rows; r++) {
        const double *wl = lds + (r & 3) * rowlen + 160 + lane;
        const int ng = n >> 2;
        if (V == 0 || V == 3) {
            for (int k4 = 0; k4 < ng; k4++) {
                const uint32_t s4 = (uint32_t)__builtin_amdgcn_readlane((int)plv, k4);
                const double *p0 = wl - (int)(s4 & 255), *p1 = wl - (int)((s4 >> 8) & 255), *p2 = wl - (int)((s4 >> 16) & 255), *p3 = wl - (int)(s4 >> 24);
                double w0[CPL], w1[CPL], w2[CPL], w3[CPL];
#pragma unroll
                for (int i = 0; i < CPL; i++) { w0[i] = ldv(p0 + i * 64); w1[i] = ldv(p1 + i * 64); w2[i] = ldv(p2 + i * 64); w3[i] = ldv(p3 + i * 64); }
                if (V == 0) {
#pragma unroll
                    for (int i = 0; i < CPL; i++) acc[i] += w0[i];
#pragma unroll
                    for (int i = 0; i < CPL; i++) acc[i] += w1[i];
#pragma unroll
                    for (int i = 0; i < CPL; i++) acc[i] += w2[i];
#pragma unroll
                    for (int i = 0; i < CPL; i++) acc[i] += w3[i];
                } else {
#pragma unroll
                    for (int i = 0; i < CPL; i++) asm volatile("" :: "v"(w0[i]), "v"(w1[i]), "v"(w2[i]), "v"(w3[i]));
                }
            }
        } else if (V == 4) {
            constexpr int GP = 8 / CPL;                         // positions per set of 8 loads
            const int ns = n / GP;
            const unsigned wlb = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) double *)wl;
            double wa[8], wb[8];
            auto ld = [&](double (&w)[8], int g) {
#pragma unroll
                for (int q = 0; q < GP; q++) {
                    const int pi = g * GP + q;
                    const uint32_t s4 = (uint32_t)__builtin_amdgcn_readlane((int)plv, pi >> 2);
                    const unsigned a = wlb - 8u * ((s4 >> (8 * (pi & 3))) & 255u);
#pragma unroll
                    for (int i = 0; i < CPL; i++) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(w[q * CPL + i]) : "v"(a), "n"(512 * i));
                }
            };
            auto ad = [&](double (&w)[8]) {
#pragma unroll
                for (int q = 0; q < GP; q++)
#pragma unroll
                    for (int i = 0; i < CPL; i++) acc[i] += w[q * CPL + i];
            };
#define WAIT(w, N) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]))
            if (ns > 0) ld(wa, 0);
            int g = 0;
            for (; g + 2 < ns; g += 2) {
                ld(wb, g + 1); WAIT(wa, 8); ad(wa);
                ld(wa, g + 2); WAIT(wb, 8); ad(wb);
            }
            if (g + 1 < ns) { ld(wb, g + 1); WAIT(wa, 8); ad(wa); WAIT(wb, 0); ad(wb); }
            else if (g < ns) { WAIT(wa, 0); ad(wa); }
        } else {
            constexpr int GP = V == 2 ? 8 : 4;
            const int ngp = n / GP;
            double wa[GP][CPL], wb[GP][CPL];
            auto ld = [&](double (&w)[GP][CPL], int g) {
#pragma unroll
                for (int q4 = 0; q4 < GP / 4; q4++) {
                    const uint32_t s4 = (uint32_t)__builtin_amdgcn_readlane((int)plv, g * (GP / 4) + q4);
#pragma unroll
                    for (int u = 0; u < 4; u++) {
                        const double *pp = wl - (int)((s4 >> (8 * u)) & 255);
#pragma unroll
                        for (int i = 0; i < CPL; i++) w[q4 * 4 + u][i] = ldv(pp + i * 64);
                    }
                }
            };
            auto ad = [&](const double (&w)[GP][CPL]) {
#pragma unroll
                for (int q = 0; q < GP; q++)
#pragma unroll
                    for (int i = 0; i < CPL; i++) acc[i] += w[q][i];
            };
            if (ngp > 0) ld(wa, 0);
            for (int g = 0; g < ngp; g += 2) {
                if (g + 1 < ngp) ld(wb, g + 1);
                ad(wa);
                if (g + 1 < ngp) { if (g + 2 < ngp) ld(wa, g + 2); ad(wb); }
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    double a = 0;
#pragma unroll
    for (int i = 0; i < CPL; i++) a += acc[i];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = a;
    if (lane == 0) atomicAdd(cyc, t1 - t0);
}

template <int CPL, int V> void run(double *d, unsigned long long *dc, int per_cu, const char *name) {
    const int rows = 4000, n = 36, rowlen = 1024, nblk = 256 * per_cu;
    const size_t ldsb = per_cu == 1 ? 150 * 1024 : 80 * 1024;
    hipFuncSetAttribute((const void *)k<CPL, V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0; unsigned long long c = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipMemset(dc, 0, 8);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<CPL, V>), dim3(nblk), dim3(256), ldsb, 0, d, dc, rows, n, rowlen);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    }
    const int gp = V == 2 ? 8 : V == 4 ? 8 / CPL : 4;
    const double la = (double)rows * (n / gp) * gp * CPL;       // load+adds per wave
    const double waves = nblk * 4.0;
    printf("%-28s CPL=%d %d WG/CU: %7.2f ms, %6.2f wave-cycles per load+add, %6.2f CU-cycles per wave-load (LDS array: 2)\n", name, CPL, per_cu, ms,
           (double)c / waves / la, ms * 1e-3 * 2.4e9 / (la * 4.0 * per_cu));
}

int main() {
    double *d; unsigned long long *dc;
    hipMalloc(&d, 512 * 256 * 8); hipMalloc(&dc, 8);
    for (int per_cu = 1; per_cu <= 2; per_cu++) {
        run<4, 0>(d, dc, per_cu, "V0 engine loop");
        run<4, 1>(d, dc, per_cu, "V1 two register sets");
        run<4, 3>(d, dc, per_cu, "V3 loads only");
        run<4, 4>(d, dc, per_cu, "V4 two sets, manual waits");
        run<2, 0>(d, dc, per_cu, "V0 engine loop");
        run<2, 1>(d, dc, per_cu, "V1 two register sets");
        run<2, 2>(d, dc, per_cu, "V2 8 positions, two sets");
        run<2, 3>(d, dc, per_cu, "V3 loads only");
        run<2, 4>(d, dc, per_cu, "V4 two sets, manual waits");
        run<1, 0>(d, dc, per_cu, "V0 engine loop");
        run<1, 2>(d, dc, per_cu, "V2 8 positions, two sets");
        run<1, 3>(d, dc, per_cu, "V3 loads only");
        run<1, 4>(d, dc, per_cu, "V4 two sets, manual waits");
    }
    return 0;
}
