// Calibration microbenchmark (not part of the product): how fast ONE workgroup on ONE compute
// unit streams a private region the size of an MH window (fh*fw*Dp doubles of residual and of
// 1/variance: 2 x 121 KiB at 128 channels), as a function of its wavefronts (NW) and of the
// 16-byte loads each lane keeps in flight (U).  A small colour launch is bound by exactly this
// ("Small colour launches").  grid = number of windows of the launch.
//   hipcc --offload-arch=gfx950 -O3 -o tools/cu_stream tools/cu_stream.hip && tools/cu_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

template <int U>
__global__ void k_stream(const double2 *__restrict__ a, const double2 *__restrict__ b, long wg_stride,
                         int n16, double *out, unsigned long long *clk) {
    // n16 = 16-byte elements of ONE stream per workgroup; both streams are read
    const double2 *pa = a + (long)blockIdx.x * wg_stride;
    const double2 *pb = b + (long)blockIdx.x * wg_stride;
    const int nt = blockDim.x;
    double2 acc = make_double2(0.0, 0.0);
    const unsigned long long t0 = wall_clock64();
    for (int i = threadIdx.x; i < n16; i += nt * U) {
        double2 va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int j = min(i + u * nt, n16 - 1);
            va[u] = pa[j];
            vb[u] = pb[j];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            acc.x = fma(va[u].x, vb[u].x, acc.x);
            acc.y = fma(va[u].y, vb[u].y, acc.y);
        }
    }
    out[(long)blockIdx.x * nt + threadIdx.x] = acc.x + acc.y;
    __syncthreads();
    if (threadIdx.x == 0) clk[blockIdx.x] = wall_clock64() - t0;
}

int main(int argc, char **argv) {
    const int Dp = argc > 1 ? atoi(argv[1]) : 128;
    const int grid = argc > 2 ? atoi(argv[2]) : 49;
    const int n16 = 121 * Dp / 2;                       // one window, one stream
    const long wg_stride = 300L * 11 * Dp / 2;          // windows of a colour class are 11 columns apart...
    const size_t elems = (size_t)wg_stride * grid + n16 + (64 << 20) / 16;
    double2 *a, *b;
    double *out;
    unsigned long long *clk;
    hipMalloc(&a, elems * 16);
    hipMalloc(&b, elems * 16);
    hipMalloc(&out, sizeof(double) * grid * 1024);
    hipMalloc(&clk, 8 * grid);
    hipMemset(a, 0, elems * 16);
    hipMemset(b, 0, elems * 16);
    // something to push the region out of the L2s between repetitions (not out of the MALL)
    double2 *junk;
    const size_t junk_bytes = 64 << 20;
    hipMalloc(&junk, junk_bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    printf("# Dp %d: window = 2 x %.1f KiB, %d workgroups\n", Dp, n16 * 16 / 1024.0, grid);
    for (int nw : {4, 8, 11, 16}) {
        for (int U : {1, 2, 4, 8}) {
            std::vector<float> ms_all;
            std::vector<unsigned long long> h(grid);
            double med_clk = 0;
            for (int rep = 0; rep < 6; ++rep) {
                hipMemsetAsync(junk, rep, junk_bytes, 0);
                hipEventRecord(e0);
                const dim3 g(grid), t(nw * 64);
                switch (U) {
                    case 1: hipLaunchKernelGGL(k_stream<1>, g, t, 0, 0, a, b, wg_stride, n16, out, clk); break;
                    case 2: hipLaunchKernelGGL(k_stream<2>, g, t, 0, 0, a, b, wg_stride, n16, out, clk); break;
                    case 4: hipLaunchKernelGGL(k_stream<4>, g, t, 0, 0, a, b, wg_stride, n16, out, clk); break;
                    default: hipLaunchKernelGGL(k_stream<8>, g, t, 0, 0, a, b, wg_stride, n16, out, clk); break;
                }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep >= 2) ms_all.push_back(ms);
                hipMemcpy(h.data(), clk, 8 * grid, hipMemcpyDeviceToHost);
                std::sort(h.begin(), h.end());
                med_clk = h[grid / 2] * 0.01;  // 100 MHz wall clock -> us
            }
            std::sort(ms_all.begin(), ms_all.end());
            const double us = ms_all[ms_all.size() / 2] * 1e3;
            printf("NW %2d  U %d : launch %6.2f us, median workgroup %5.2f us  -> %6.1f GB/s per CU\n", nw, U,
                   us, med_clk, 2.0 * n16 * 16 / (med_clk * 1e-6) / 1e9);
        }
    }
    return 0;
}
