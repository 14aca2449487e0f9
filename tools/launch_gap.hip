// What does a kernel boundary cost on this part, and what changes it?  (DESIGN.md section 3:
// a colour launch that does not fill the chip pays 1.3-2.4 us of boundary, 121 times a sweep.)
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/launch_gap tools/launch_gap.hip && tools/launch_gap
//
// Chains of N dependent launches in one stream, 64 workgroups x 320 threads each, a few
// hundred cycles of work; period = wall time / N.  Variants:
//   plain        no scratch, 16-byte argument
//   bigarg       a 640-byte by-value argument struct (MHArgs is ~600 bytes)
//   scratch      a private array the compiler cannot remove (80 bytes per lane reserved)
//   call         a non-inlined device function (stack reserved, never spilled to)
//   graph        the plain chain captured once into a hipGraph and replayed
//   storing      every thread stores 16 bytes (dirty lines to write back at the boundary)
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                    \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

struct Big {
    double v[80];
};

__global__ void k_plain(double *p, int n) {
    if (n < 0) p[threadIdx.x] = 1.0;
}
__global__ void k_bigarg(Big b, double *p, int n) {
    if (n < 0) p[threadIdx.x] = b.v[n & 63];
}
__global__ void k_scratch(double *p, int n) {
    volatile double a[10];
    if (n < 0) {
        for (int i = 0; i < 10; ++i) a[i] = p[i];
        p[threadIdx.x] = a[n & 7];
    }
}
__device__ __noinline__ double callee(double x, int n) {
    for (int i = 0; i < n; ++i) x = x * 1.0000001 + 1e-9;
    return x;
}
__global__ void k_call(double *p, int n) {
    if (n < 0) p[threadIdx.x] = callee(p[0], -n);
}
__global__ void k_storing(double *p, int n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    reinterpret_cast<double2 *>(p)[i] = make_double2((double)n, 1.0);
}

// ~`ticks` x 10 ns of "work": every workgroup spins on the 100 MHz wall clock, then stores
__global__ void k_work(double *p, int n, int ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(1);
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    reinterpret_cast<double2 *>(p)[i] = make_double2((double)n, 1.0);
}

template <class F>
static double chain_us(hipStream_t st, int n, F launch) {
    for (int i = 0; i < 200; ++i) launch(i);
    (void)hipStreamSynchronize(st);
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) launch(i);
    (void)hipStreamSynchronize(st);
    const auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
}

int main() {
    hipStream_t st;
    CK(hipStreamCreate(&st));
    double *buf;
    CK(hipMalloc(&buf, 64 * 320 * 16 + 4096));
    const dim3 grid(64), block(320);
    const int N = 4000;
    Big big = {};
    printf("period of a chain of %d dependent launches (64 x 320 threads), us per launch\n", N);
    printf("  plain    %.2f\n", chain_us(st, N, [&](int i) { hipLaunchKernelGGL(k_plain, grid, block, 0, st, buf, i); }));
    printf("  bigarg   %.2f\n", chain_us(st, N, [&](int i) { hipLaunchKernelGGL(k_bigarg, grid, block, 0, st, big, buf, i); }));
    printf("  scratch  %.2f\n", chain_us(st, N, [&](int i) { hipLaunchKernelGGL(k_scratch, grid, block, 0, st, buf, i); }));
    printf("  call     %.2f\n", chain_us(st, N, [&](int i) { hipLaunchKernelGGL(k_call, grid, block, 0, st, buf, i); }));
    printf("  storing  %.2f\n", chain_us(st, N, [&](int i) { hipLaunchKernelGGL(k_storing, grid, block, 0, st, buf, i); }));
    printf("  plain, 30 KB of dynamic LDS  %.2f\n",
           chain_us(st, N, [&](int i) { hipLaunchKernelGGL(k_plain, grid, block, 30 * 1024, st, buf, i); }));
    // the same chain as a graph: 121 kernel nodes, replayed
    hipGraph_t graph;
    hipGraphExec_t exec;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < 121; ++i) hipLaunchKernelGGL(k_plain, grid, block, 0, st, buf, i);
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(exec, st));
    CK(hipStreamSynchronize(st));
    {
        const int R = 40;
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < R; ++r) CK(hipGraphLaunch(exec, st));
        CK(hipStreamSynchronize(st));
        const auto t1 = std::chrono::steady_clock::now();
        printf("  graph    %.2f   (121 kernel nodes per graph, %d replays)\n",
               std::chrono::duration<double, std::micro>(t1 - t0).count() / (R * 121), R);
    }
    CK(hipGraphExecDestroy(exec));
    CK(hipGraphDestroy(graph));
    // kernels that take ~8 us each (the CPU is never the bottleneck then): what is left of the
    // period is the boundary itself -- stream launches against a replayed graph
    for (int ticks : {400, 800}) {
        const double s_us = chain_us(st, 2000, [&](int i) { hipLaunchKernelGGL(k_work, grid, block, 0, st, buf, i, ticks); });
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 121; ++i) hipLaunchKernelGGL(k_work, grid, block, 0, st, buf, i, ticks);
        CK(hipStreamEndCapture(st, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(exec, st));
        CK(hipStreamSynchronize(st));
        const int R = 20;
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < R; ++r) CK(hipGraphLaunch(exec, st));
        CK(hipStreamSynchronize(st));
        const auto t1 = std::chrono::steady_clock::now();
        const double g_us = std::chrono::duration<double, std::micro>(t1 - t0).count() / (R * 121);
        printf("  work %4.1f us per kernel: stream %.2f, graph %.2f us per launch (boundary %.2f / %.2f)\n",
               ticks * 0.01, s_us, g_us, s_us - ticks * 0.01, g_us - ticks * 0.01);
        CK(hipGraphExecDestroy(exec));
        CK(hipGraphDestroy(graph));
    }
    CK(hipFree(buf));
    return 0;
}
