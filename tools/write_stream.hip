// Calibration microbenchmark (not part of the product): how fast the chip WRITES a cube that
// nobody reads in the same launch -- the bound of k_lines / k_lines_dense (the line cube: one
// exp per voxel, 92 MB out, nothing in).  Variants: plain 16-byte stores
// with one wavefront per KiB (k_lines' shape) or a grid-stride loop, non-temporal stores, and a
// read-only pass and a copy of the same size beside them.
//   hipcc --offload-arch=gfx950 -O3 -o tools/write_stream tools/write_stream.hip && tools/write_stream [MB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void k_write(double2 *out, long n16, double v) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x)
        out[i] = make_double2(v, v + 1.0);
}
__global__ void k_write_nt(double2 *out, long n16, double v) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) {
        __builtin_nontemporal_store(v, &out[i].x);
        __builtin_nontemporal_store(v + 1.0, &out[i].y);
    }
}
__global__ void k_read(const double2 *in, long n16, double *sink) {
    double acc = 0.0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) {
        const double2 v = in[i];
        acc += v.x + v.y;
    }
    if (acc == 12345.678) *sink = acc;
}
__global__ void k_copy(const double2 *in, double2 *out, long n16) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x)
        out[i] = in[i];
}

template <class F>
static double timed(F launch, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3 / reps;
}

int main(int argc, char **argv) {
    const double mb = argc > 1 ? atof(argv[1]) : 92.16;
    const long n16 = (long)(mb * 1e6 / 16);
    double2 *a, *b;
    double *sink;
    hipMalloc(&a, n16 * 16);
    hipMalloc(&b, n16 * 16);
    hipMalloc(&sink, 8);
    hipMemset(a, 0, n16 * 16);
    hipMemset(b, 0, n16 * 16);
    const int reps = 50;
    const unsigned full = (unsigned)((n16 + 255) / 256);
    printf("%.2f MB per pass (us per pass, TB/s of the bytes the pass names)\n", mb);
    struct { const char *name; unsigned grid; } shapes[] = {{"one 16-byte store per thread", full},
                                                             {"grid-stride, 256 x 16 workgroups", 256 * 16},
                                                             {"grid-stride, 256 x 4 workgroups", 256 * 4}};
    for (auto &sh : shapes) {
        double us = timed([&] { hipLaunchKernelGGL(k_write, dim3(sh.grid), dim3(256), 0, 0, a, n16, 1.0); }, reps);
        printf("  write      %-36s %7.1f us  %.2f TB/s\n", sh.name, us, mb / us);
        us = timed([&] { hipLaunchKernelGGL(k_write_nt, dim3(sh.grid), dim3(256), 0, 0, a, n16, 1.0); }, reps);
        printf("  write, nt  %-36s %7.1f us  %.2f TB/s\n", sh.name, us, mb / us);
        us = timed([&] { hipLaunchKernelGGL(k_read, dim3(sh.grid), dim3(256), 0, 0, a, n16, sink); }, reps);
        printf("  read       %-36s %7.1f us  %.2f TB/s\n", sh.name, us, mb / us);
        us = timed([&] { hipLaunchKernelGGL(k_copy, dim3(sh.grid), dim3(256), 0, 0, a, b, n16); }, reps);
        printf("  copy       %-36s %7.1f us  %.2f TB/s (read + write)\n", sh.name, us, 2 * mb / us);
    }
    double us = timed([&] { hipMemsetAsync(a, 0, n16 * 16, 0); }, reps);
    printf("  hipMemsetAsync %40.1f us  %.2f TB/s\n", us, mb / us);
    return 0;
}
