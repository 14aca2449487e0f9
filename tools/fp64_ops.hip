// Calibration microbenchmark (not part of the product): issue interval of the fp64 vector
// instructions the kernels of this library are made of, on ONE wavefront: 16 independent
// instructions of one kind, repeated, timed with s_memtime (shader clock).  The line kernel, the
// convolution and the tails of the sweep kernels are bound by vector issue:
// this says what an instruction costs.
//   hipcc --offload-arch=gfx950 -O3 -o tools/fp64_ops tools/fp64_ops.hip && tools/fp64_ops
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ void k_op(double *out, unsigned long long *cyc, int iters, double seed) {
    double r[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) r[k] = seed + k * 0.001 + threadIdx.x * 1e-6;
    const double c = 0.999999, d = 1e-9;
    int e = 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#define FMA(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(r[k]) : "v"(c), "v"(d));
#define MUL(k) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(r[k]) : "v"(c));
#define ADD(k) asm volatile("v_add_f64 %0, %0, %1" : "+v"(r[k]) : "v"(d));
#define LDEXP(k) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(r[k]) : "v"(e));
#define RNDNE(k) asm volatile("v_rndne_f64 %0, %0" : "+v"(r[k]));
#define RCP(k) asm volatile("v_rcp_f64 %0, %0" : "+v"(r[k]));
#define RSQ(k) asm volatile("v_rsq_f64 %0, %0" : "+v"(r[k]));
#define SQRT(k) asm volatile("v_sqrt_f64 %0, %0" : "+v"(r[k]));
#define CVTI(k) { int t; asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(t) : "v"(r[k])); e ^= t & 0; }
#define DIVFIX(k) asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(r[k]) : "v"(c), "v"(d));
#define FMA32(k) { float f = (float)k; asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f)); e ^= ((int)f) & 0; }
#define MOV64(k) asm volatile("v_mov_b64 %0, %1" : "+v"(r[k]) : "v"(c));
#define FRACT(k) asm volatile("v_fract_f64 %0, %0" : "+v"(r[k]));
#define MAXF(k) asm volatile("v_max_f64 %0, %0, %1" : "+v"(r[k]) : "v"(d));
        if (OP == 0) { REP16(FMA) }
        if (OP == 1) { REP16(MUL) }
        if (OP == 2) { REP16(ADD) }
        if (OP == 3) { REP16(LDEXP) }
        if (OP == 4) { REP16(RNDNE) }
        if (OP == 5) { REP16(RCP) }
        if (OP == 6) { REP16(RSQ) }
        if (OP == 7) { REP16(SQRT) }
        if (OP == 8) { REP16(CVTI) }
        if (OP == 9) { REP16(DIVFIX) }
        if (OP == 10) { REP16(FMA32) }
        if (OP == 11) { REP16(MOV64) }
        if (OP == 12) { REP16(FRACT) }
        if (OP == 13) { REP16(MAXF) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += r[k];
    out[threadIdx.x] = s + e;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int OP>
static void run(const char *name) {
    double *out;
    unsigned long long *cyc, h = 0;
    hipMalloc(&out, 64 * 8);
    hipMalloc(&cyc, 8);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_op<OP>, dim3(1), dim3(64), 0, 0, out, cyc, iters, 1.0);
        hipDeviceSynchronize();
    }
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("  %-18s %6.2f cycles per instruction (one wavefront, 16 independent in a row)\n", name,
           (double)h / (iters * 16.0));
    hipFree(out);
    hipFree(cyc);
}

int main() {
    printf("issue interval on one wavefront of 64 (shader-clock cycles, s_memtime):\n");
    run<0>("v_fma_f64");
    run<1>("v_mul_f64");
    run<2>("v_add_f64");
    run<13>("v_max_f64");
    run<11>("v_mov_b64");
    run<3>("v_ldexp_f64");
    run<4>("v_rndne_f64");
    run<12>("v_fract_f64");
    run<8>("v_cvt_i32_f64");
    run<5>("v_rcp_f64");
    run<6>("v_rsq_f64");
    run<7>("v_sqrt_f64");
    run<9>("v_div_fixup_f64");
    run<10>("v_fma_f32");
    return 0;
}
