// Micro-benchmark: sustained FP64 vector issue rate of one MI355X for independent v_mul_f64 / v_add_f64 / v_fma_f64 streams
// (24 independent accumulators per lane, no memory traffic), at one and two waves per SIMD.  What fraction of
// 256 CU x 4 SIMD x 16 lanes x 2.4 GHz = 39.3 T instr-lanes/s does plain FP64 code reach?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int V>
__global__ void __launch_bounds__(512) k(double *out, int iters, double c, double d) {
    double p[24];
#pragma unroll
    for (int s = 0; s < 24; s++) p[s] = 1.0 + 1e-3 * s + threadIdx.x * 1e-9;
    const double cv = c + threadIdx.x * 1e-18;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int s = 0; s < 24; s++) {
            if (V == 0) p[s] = p[s] * c;
            else if (V == 1) p[s] = p[s] + d;
            else if (V == 2) p[s] = __builtin_fma(p[s], c, d);
            else if (V == 3) { p[s] = p[s] * c; p[s] = p[s] + d; }      // mul then dependent add (two instructions)
            else if (V == 4) p[s] = p[s] * cv;                           // multiplier in a VGPR instead of an SGPR pair
            else if (V == 5) { if (s & 1) p[s] = p[s] + d; else p[s] = p[s] * c; }   // independent multiplies and adds alternating
            else if (V == 6) { p[s] = p[s] * c; }                        // (second round below: 48 multiplies per loop trip)
        }
        if (V == 6) {
#pragma unroll
            for (int s = 0; s < 24; s++) p[s] = p[s] * c;
        }
    }
    double a = 0;
#pragma unroll
    for (int s = 0; s < 24; s++) a += p[s];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = a;
}
template <int V> void run(double *d, int threads, const char *name) {
    const int iters = 20000, nblk = 256;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<V>, dim3(nblk), dim3(threads), 0, 0, d, iters, 0.9999999, 1e-12);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    const double ops = (double)nblk * threads * 24.0 * iters * (V == 3 || V == 6 ? 2 : 1);
    printf("%-22s threads %3d: %.3f ms, %.2f T instr-lanes/s = %.1f %% of 39.3\n", name, threads, ms, ops / ms * 1e-9, ops / ms * 1e-9 / 39.3 * 100);
}
int main() {
    double *d; hipMalloc(&d, 256 * 512 * 8);
    for (int threads : {256, 512}) {
        run<0>(d, threads, "v_mul_f64");
        run<1>(d, threads, "v_add_f64");
        run<2>(d, threads, "v_fma_f64");
        run<3>(d, threads, "mul + dependent add");
        run<4>(d, threads, "v_mul_f64, VGPR factor");
        run<5>(d, threads, "mul / add independent");
        run<6>(d, threads, "v_mul_f64 x48 per trip");
    }
    return 0;
}
