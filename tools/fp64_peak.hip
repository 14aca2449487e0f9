// Calibration microbenchmark (not part of the product): sustained rate of
// independent v_fma_f64 chains, all CUs busy, NW waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int NACC>
__global__ __launch_bounds__(256) void k_fma(double *out, int iters, double a, double b) {
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = fma(acc[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main(int argc, char **argv) {
    const int blocks_per_cu = argc > 1 ? atoi(argv[1]) : 2;
    const int iters = 20000, NACC = 16;
    const int grid = 256 * blocks_per_cu;
    double *d;
    hipMalloc(&d, sizeof(double) * grid * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_fma<NACC>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0000001, 1e-9);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flops = 2.0 * NACC * iters * (double)grid * 256;
        printf("blocks/CU %d: %.3f ms  %.1f TFLOP/s fp64  (%.2f cycles per wave-FMA per SIMD at 2.4 GHz)\n",
               blocks_per_cu, ms, flops / ms / 1e9,
               2.4e9 * ms * 1e-3 / ((double)NACC * iters * blocks_per_cu));
    }
    return 0;
}
