// One-pass separable convolution LSF (x) FSF for gfx950: the FSF stencil of
// lib/run.py:1027-1029 (scipy convolve2d 'same', zero boundary) and the LSF pass
// of lib/convolution.py:89-120 (closed form, power-of-two depth) in ONE trip
// through HBM -- cube in, cube out, nothing in between.
//
// Why another kernel than k_spatial_march (d3d_kernels.h): that one keeps a
// 3-column register ring per thread (~200 VGPRs, two wavefronts per SIMD), loads
// its input rows itself and needs a second pass for the LSF.  Measured there:
// fp64 issue at two wavefronts per SIMD tops out at ~41 TFLOP/s, the row loads
// cost issue slots, and the LSF pass costs a full extra HBM round trip
// (DESIGN.md section 3).  Here
//   * a workgroup is NW compute wavefronts + ONE loader wavefront.  The loader
//     streams input rows into a 3-deep LDS ring with LDS-DMA (global_load_lds,
//     no VGPR destination, two rows in flight) -- the compute waves never issue
//     a global load for the stencil, so their registers hold only the ring;
//   * a compute wavefront owns ONE output column (its 128-channel spectrum: lane
//     <-> z-pair) and marches down the strip with an FS-slot register ring of
//     pending output rows (44 VGPRs at FS = 11): <= 128 VGPRs per thread, four
//     wavefronts per SIMD, where v_fma_f64 issues at 52-58 TFLOP/s instead of 41;
//   * x- and y-mirror symmetry of the FSF folded as in the march kernel
//     ((FHH+1)^2 FMAs + FS adds per output and input row), taps as scalar
//     operands (SGPRs) -- no LDS or VGPR traffic for them;
//   * the finished output row goes through a wave-private LDS spectrum buffer
//     and the dense LSF taps before it is stored (optionally as data - conv).
// Zero boundary: out-of-range columns are zero slots in LDS (never loaded),
// out-of-range rows are skipped steps.  The march always runs top-down, so an
// output's summation order does not depend on where its strip starts: a tile of
// a cube reproduces the full cube's values bit for bit.
#pragma once
#include <hip/hip_runtime.h>

namespace d3d {

struct ConvRowsArgs {
    int H, W, HY;        // cube rows / columns, output rows per strip
    int ngx, ngy;        // column groups, row strips
    const double *quad;  // [(FHH+1)^2] quadrant taps: quad[a*(FHH+1)+m] = fsf[FHH-a][m]
    const double *wl;    // dense LSF weights [2*LSF_RL+1] (used when LSF)
    const double *data;  // residual epilogue: out = data - conv (used when RESID)
    int xcd_remap;
};

constexpr int CONV_DP = 128;  // doubles per spectrum: one wavefront of z-pairs
constexpr int CONV_NBUF = 3;  // LDS ring of input rows

template <int FS, int NW>
__host__ __device__ constexpr size_t conv_rows_lds_bytes() {
    return ((size_t)CONV_NBUF * (NW + FS - 1) * CONV_DP + (size_t)NW * (CONV_DP + 2 * LSF_RL) +
            CONV_DP) * sizeof(double);
}

__device__ __forceinline__ void conv_glds16(const double *gsrc, double *lds_dst) {
    // 64 lanes x 16 B: per-lane global source, wave-uniform LDS base + lane*16
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void *)gsrc,
        (__attribute__((address_space(3))) void *)lds_dst, 16, 0, 0);
}

template <int FS, int NW, bool LSF, bool LSYM, bool RESID>
__global__ __launch_bounds__((NW + 1) * 64) void k_conv_rows(ConvRowsArgs A,
                                                              const double *__restrict__ in,
                                                              double *__restrict__ out) {
    constexpr int FHH = (FS - 1) / 2, NQ = FHH + 1, NC = NW + FS - 1, DP = CONV_DP;
    constexpr int NBUF = CONV_NBUF, RL = LSF_RL;
    extern __shared__ double smem[];
    double *rows = smem;                                // [NBUF][NC][DP]
    double *spec = rows + (size_t)NBUF * NC * DP;       // [NW][DP + 2 RL]
    double *dummy = spec + (size_t)NW * (DP + 2 * RL);  // [DP]: target of out-of-range loads
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    // XCD-aware block order: consecutive LOGICAL blocks (x-neighbours of one row
    // strip, FS-1 common input columns) on one XCD and its L2 (bijective remap)
    int blk = blockIdx.x;
    if (A.xcd_remap) {
        const int nb = gridDim.x, q = nb / 8, rm = nb % 8, xcd = blk % 8;
        blk = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + blk / 8;
    }
    const int gy = blk / A.ngx, gx = blk - gy * A.ngx;
    const int x0 = gx * NW, y0 = gy * A.HY;
    const int yend = min(y0 + A.HY, A.H);
    const int nsteps = (yend - y0) + 2 * FHH;
    const long rowstride = (long)A.W * DP;

    if (wave == NW) {
        // ---- loader wavefront ---------------------------------------------------
        for (int c = 0; c < NC; ++c) {
            const int xx = x0 - FHH + c;
            if (xx < 0 || xx >= A.W)
                for (int b = 0; b < NBUF; ++b)
                    *reinterpret_cast<double2 *>(rows + ((size_t)b * NC + c) * DP + 2 * lane) =
                        make_double2(0.0, 0.0);
        }
        auto issue = [&](int i) -> bool {
            const int r = y0 - FHH + i;
            if (i >= nsteps || r < 0 || r >= A.H) return false;
            double *dst = rows + (size_t)(i % NBUF) * NC * DP;
            const double *src = in + (long)r * rowstride + 2 * lane;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int xx = x0 - FHH + c;
                const bool ok = xx >= 0 && xx < A.W;
                // always NC loads per row: the counted wait below relies on it
                conv_glds16(src + (long)(ok ? xx : 0) * DP, ok ? dst + (size_t)c * DP : dummy);
            }
            return true;
        };
        issue(0);
        bool next_issued = issue(1);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // zero slots written
        for (int i = 0; i < nsteps; ++i) {
            // row i has landed when at most row i+1's loads are outstanding
            if (next_issued)
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NC) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // B_i: row i is readable; the buffer of row i-1 is free again
            asm volatile("s_barrier" ::: "memory");
            next_issued = issue(i + 2);
        }
        return;
    }

    // ---- compute wavefront: output column x0 + wave --------------------------------
    const int x = x0 + wave;
    const bool col_ok = x < A.W;
    double2 ring[FS];
#pragma unroll
    for (int k = 0; k < FS; ++k) ring[k] = make_double2(0.0, 0.0);
    double *myspec = spec + (size_t)wave * (DP + 2 * RL);
    double2 dnext = make_double2(0.0, 0.0);  // RESID: data of the row finished next
    asm volatile("s_barrier" ::: "memory");  // prologue
    for (int sbase = 0; sbase < nsteps; sbase += FS) {
#pragma unroll
        for (int ph = 0; ph < FS; ++ph) {
            const int i = sbase + ph;
            if (i < nsteps) {
                // every LDS read of the previous step has been consumed (data dependences)
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // B_i
                const int r = y0 - FHH + i;
                const int oy = r - FHH;  // the output row this step finishes
                double2 dcur = dnext;
                if constexpr (RESID) {
                    // one step ahead: the data row of the output finished by step i+1
                    const int oyn = oy + 1;
                    if (col_ok && oyn >= y0 && oyn < yend)
                        dnext = *reinterpret_cast<const double2 *>(A.data + (long)oyn * rowstride +
                                                                   (long)x * DP + 2 * lane);
                }
                if (col_ok && r >= 0 && r < A.H) {
                    const double *rb = rows + ((size_t)(i % NBUF) * NC + wave) * DP + 2 * lane;
                    double2 P[NQ];
#pragma unroll
                    for (int m = 0; m < FHH; ++m) {
                        const double2 lo = *reinterpret_cast<const double2 *>(rb + (size_t)m * DP);
                        const double2 hi =
                            *reinterpret_cast<const double2 *>(rb + (size_t)(FS - 1 - m) * DP);
                        P[m].x = lo.x + hi.x;
                        P[m].y = lo.y + hi.y;
                    }
                    P[FHH] = *reinterpret_cast<const double2 *>(rb + (size_t)FHH * DP);
#pragma unroll
                    for (int a = 0; a <= FHH; ++a) {
                        const int ylo = r - a, yhi = r + a;  // output rows of slots FHH-a, FHH+a
                        const bool lo_ok = ylo >= y0 && ylo < yend;
                        const bool hi_ok = a > 0 && yhi >= y0 && yhi < yend;
                        if (lo_ok || hi_ok) {
                            double2 T;
                            {
                                const double q0 = A.quad[a * NQ];
                                T.x = q0 * P[0].x;
                                T.y = q0 * P[0].y;
                            }
#pragma unroll
                            for (int m = 1; m < NQ; ++m) {
                                const double q = A.quad[a * NQ + m];
                                T.x = fma(q, P[m].x, T.x);
                                T.y = fma(q, P[m].y, T.y);
                            }
                            if (lo_ok) {
                                ring[(ph + FHH - a) % FS].x += T.x;
                                ring[(ph + FHH - a) % FS].y += T.y;
                            }
                            if (hi_ok) {
                                ring[(ph + FHH + a) % FS].x += T.x;
                                ring[(ph + FHH + a) % FS].y += T.y;
                            }
                        }
                    }
                }
                if (col_ok && oy >= y0 && oy < yend) {
                    double2 v = ring[ph % FS];
                    if constexpr (LSF) {
                        // LSF on the finished row: spectrum -> wave-private LDS buffer with a
                        // circular halo of RL channels -> aligned 16-byte window reads
                        *reinterpret_cast<double2 *>(myspec + RL + 2 * lane) = v;
                        if (2 * lane < RL)
                            *reinterpret_cast<double2 *>(myspec + DP + RL + 2 * lane) = v;
                        if (2 * lane >= DP - RL)
                            *reinterpret_cast<double2 *>(myspec + RL + 2 * lane - DP) = v;
                        __builtin_amdgcn_wave_barrier();  // LDS is in order per wavefront
                        const double *bt = myspec + 2 * lane;
                        double2 acc = make_double2(0.0, 0.0);
#pragma unroll
                        for (int j = 0; j < RL + 1; ++j) {
                            const double2 p = *reinterpret_cast<const double2 *>(bt + 2 * j);
                            // p = (w[2j], w[2j+1]); acc.x = sum wl[k] w[k], acc.y = sum wl[k] w[k+1]
                            auto wgt = [&](int k) -> double {
                                return A.wl[LSYM && k > RL ? 2 * RL - k : k];
                            };
                            if (2 * j <= 2 * RL) acc.x = fma(wgt(2 * j), p.x, acc.x);
                            if (2 * j + 1 <= 2 * RL) acc.x = fma(wgt(2 * j + 1), p.y, acc.x);
                            if (2 * j - 1 >= 0) acc.y = fma(wgt(2 * j - 1), p.x, acc.y);
                            if (2 * j <= 2 * RL) acc.y = fma(wgt(2 * j), p.y, acc.y);
                        }
                        __builtin_amdgcn_wave_barrier();
                        v = acc;
                    }
                    if constexpr (RESID) {
                        v.x = dcur.x - v.x;
                        v.y = dcur.y - v.y;
                    }
                    *reinterpret_cast<double2 *>(out + (long)oy * rowstride + (long)x * DP +
                                                 2 * lane) = v;
                }
                ring[ph % FS] = make_double2(0.0, 0.0);
            }
        }
    }
}

}  // namespace d3d
