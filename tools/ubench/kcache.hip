// Micro-benchmark: scalar-cache capacity as seen by one workgroup per CU streaming a window of S bytes with s_load_dwordx16
// (all waves of the workgroup read the same addresses, different workgroups different windows).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef const uint32_t __attribute__((address_space(4))) *cu32p;
__global__ void __launch_bounds__(512) k(const uint32_t *buf, int window_dw, int iters, unsigned *out, long long *cyc) {
    cu32p p = (cu32p)(buf + (size_t)blockIdx.x * window_dw);
    unsigned acc = 0;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++)
        for (int o = 0; o < window_dw; o += 16) {
            uint32_t v[16];
#pragma unroll
            for (int q = 0; q < 16; q++) v[q] = p[o + q];
#pragma unroll
            for (int q = 0; q < 16; q++) acc += v[q];
        }
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { out[blockIdx.x] = acc; cyc[blockIdx.x] = t1 - t0; }
}
int main() {
    const int nblk = 256;
    uint32_t *d; unsigned *o; long long *c;
    hipMalloc(&d, (size_t)nblk * 65536 * 4 + 64); hipMemset(d, 1, (size_t)nblk * 65536 * 4 + 64); hipMalloc(&o, nblk * 4); hipMalloc(&c, nblk * 8);
    for (int threads : {64, 512})
    for (int kb : {1, 2, 4, 6, 8, 10, 12, 14, 16, 20, 24, 32, 64, 128}) {
        int wdw = kb * 256; int iters = 4096 / kb; if (iters < 4) iters = 4;
        hipLaunchKernelGGL(k, dim3(nblk), dim3(threads), 0, 0, d, wdw, 2, o, c); hipDeviceSynchronize();
        hipLaunchKernelGGL(k, dim3(nblk), dim3(threads), 0, 0, d, wdw, iters, o, c); hipDeviceSynchronize();
        std::vector<long long> h(nblk); hipMemcpy(h.data(), c, nblk * 8, hipMemcpyDeviceToHost);
        double m = 0; for (auto x : h) m += x; m /= nblk;
        printf("threads %3d window %3d KB: %.1f cycles per s_load_dwordx16 (64 B)\n", threads, kb, m / ((double)iters * wdw / 16));
    }
    return 0;
}
