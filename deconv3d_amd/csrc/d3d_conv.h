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
    int H, W, HY;  // cube rows / columns, output rows per strip
    int ngx, ngy;  // column groups, row strips
    int xcd_remap;
    int Dp;        // ZB: the cube's padded depth = column stride in doubles (any even number)
};
// The tap tables come as separate `const double *__restrict__` kernel arguments (not
// inside the struct): only then does the compiler know that nothing this kernel stores
// can alias them, loads them with scalar loads ONCE, and keeps them in SGPRs for the
// whole march (v_fma_f64 takes one scalar operand).  Through a struct member they were
// re-fetched with vector global loads in every step, each a full memory latency on the
// critical path (measured: 107 us per cube instead of 5x).
//   quad  [(FHH+1)^2] quadrant taps by distance from the centre tap (outer-product FSFs:
//         row 0 = v[e], row 1 = u[a] with fsf[FHH-a][FHH-e] = u[a] v[e]),
//         quad[a*(FHH+1)+e] = fsf[FHH-a][FHH-e]; TSYM: the FSF is also symmetric under
//         transposition (quad[a][e] == quad[e][a]: every radial FSF), only e <= a is read
//   wl    dense LSF weights [2*LSF_RL+1] (LSF); LSYM: mirror-symmetric, first RL+1 read
//   data  residual epilogue out = data - conv (RESID)

constexpr int CONV_DP = 128;  // doubles per wavefront: 64 z-pairs
constexpr int CONV_NBUF = 4;  // LDS ring of input rows: two PAIRS (one read, one in flight)

// DPS = doubles per spectrum (the cube's padded depth): 128 -- one wavefront, one output
// column -- or 64 / 32: a wavefront owns SPW = 128 / DPS ADJACENT output columns, lane group
// g = lane / (DPS/2) the column x + g.  Adjacent spectra are adjacent in memory and in the
// LDS rows, so a wavefront's 1 KiB accesses stay contiguous and the stencil's column
// offsets become multiples of DPS doubles instead of 128.
//
// ZB (round 3, any depth above 128, FSF only): the spectrum is cut into Z-BLOCKS of 128
// channels and a workgroup takes one block of its column group (grid = z-blocks x row strips
// x column groups).  The FSF acts channel by channel, so the blocks are independent; the
// kernel is the 128-channel one with the column stride (A.Dp doubles) and the block's channel
// offset as run-time numbers, and a lane predicate for the ragged last block (a lane whose
// z-pair lies beyond Dp never loads -- its LDS slots keep the zero written at the start -- and
// never stores).  The LSF couples channels across blocks: it is applied by the line kernel
// (forward model) or by its own pass (an arbitrary cube), never here.
template <int FS, int NW, int DPS>
struct ConvGeo {
    static constexpr int SPW = CONV_DP / DPS;                        // spectra per wavefront
    static constexpr int NWC = NW * SPW;                             // output columns per workgroup
    static constexpr int RLEN = (NWC + FS - 1) * DPS;                // doubles of an input row segment
    static constexpr int NCH = (RLEN + CONV_DP - 1) / CONV_DP;       // 1 KiB chunks the loader moves
    static constexpr int RBUF = NCH * CONV_DP;                       // doubles per LDS row buffer
    static constexpr int SPEC = SPW * (DPS + 2 * LSF_RL);            // LSF scratch per wavefront
};

template <int FS, int NW, int DPS = CONV_DP>
__host__ __device__ constexpr size_t conv_rows_lds_bytes() {
    return ((size_t)CONV_NBUF * ConvGeo<FS, NW, DPS>::RBUF + (size_t)NW * ConvGeo<FS, NW, DPS>::SPEC +
            CONV_DP) * sizeof(double);
}

__device__ __forceinline__ void conv_glds16(const double *gsrc, double *lds_dst) {
    // 64 lanes x 16 B: per-lane global source, wave-uniform LDS base + lane*16
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void *)gsrc,
        (__attribute__((address_space(3))) void *)lds_dst, 16, 0, 0);
}

// One step of a compute wavefront: input row r = y0 - FHH + i from LDS into the ring,
// output row r - FHH out.  PH = i mod FS (static ring indices).
// Range tests are kept to ONE bit test per tap row: a contribution to an output row
// outside the strip lands in a ring slot that is never stored (the slot of a row above
// the strip is overwritten when it turns into the newest slot, rows below never
// finish), so the two slot updates of a tap row need no test of their own, and an
// input row outside the cube is a row of zeros the loader wrote.  `need` has bit a set
// when tap rows FHH-a / FHH+a reach an output row of the strip at all: that skips the
// dot products of the first and last FHH steps (10 % of a strip's FMAs) for one scalar
// bit test each.
template <int FS, int NW, bool LSF, bool RESID, bool SEP, int PH, int DPS, bool ZB>
__device__ __forceinline__ void conv_rows_step(int i, int y0, int yend, long rowstride, int cstride,
                                               int x, bool xok, int wave, int lane, const double *rows,
                                               double *myspec,
                                               const double (&q)[(FS + 1) / 2][(FS + 1) / 2],
                                               const double (&w)[2 * LSF_RL + 1],
                                               const double *__restrict__ data,
                                               double *__restrict__ out, double2 (&ring)[FS],
                                               double2 &dnext) {
    constexpr int FHH = (FS - 1) / 2, NQ = FHH + 1, DP = DPS;
    constexpr int NBUF = CONV_NBUF, RL = LSF_RL;
    constexpr int RBUF = ConvGeo<FS, NW, DPS>::RBUF, HLS = DPS / 2;
    const int zl = lane % HLS;        // z-pair within the spectrum (lane group g = lane / HLS)
    const int oy = y0 - 2 * FHH + i;  // the output row this step finishes
    const bool store = oy >= y0;      // (oy < yend always: i < nsteps)
    // tap rows at distance a from the centre reach output rows r-a and r+a, r = y0-FHH+i:
    // inside the strip iff  0 <= i-FHH-a < n  or  0 <= i-FHH+a < n
    const int n = yend - y0, d = i - FHH;
    unsigned need = 0;
#pragma unroll
    for (int a = 0; a <= FHH; ++a)
        need |= (unsigned)(((unsigned)(d - a) < (unsigned)n) || ((unsigned)(d + a) < (unsigned)n)) << a;
    const double2 dcur = dnext;
    if constexpr (RESID) {
        // one step ahead: the data row of the output finished by step i+1
        if (xok && oy + 1 >= y0 && oy + 1 < yend)
            dnext = *reinterpret_cast<const double2 *>(data + (long)(oy + 1) * rowstride +
                                                       (long)x * (ZB ? cstride : DP) + 2 * zl);
    }
    {
        const double *rb = rows + (size_t)(i % NBUF) * RBUF + (size_t)wave * CONV_DP + 2 * lane;
        // P[e]: the two inputs at distance e from the output's column, folded
        double2 P[NQ];
        P[0] = *reinterpret_cast<const double2 *>(rb + (size_t)FHH * DP);
#pragma unroll
        for (int e = 1; e <= FHH; ++e) {
            const double2 lo = *reinterpret_cast<const double2 *>(rb + (size_t)(FHH - e) * DP);
            const double2 hi = *reinterpret_cast<const double2 *>(rb + (size_t)(FHH + e) * DP);
            P[e].x = lo.x + hi.x;
            P[e].y = lo.y + hi.y;
        }
        if constexpr (SEP) {
            // outer-product FSF, fsf[k][m] = u[k] v[m] (every Gaussian with pa = 0): reduce the
            // row along x once, X = sum_e v[e] P[e], and feed the slots with u[a] X --
            // 2 (FHH+1) + FS FMAs per output instead of (FHH+1)^2 + FS.  q[0][e] = v, q[1][a] = u.
            double2 X;
            X.x = q[0][0] * P[0].x;
            X.y = q[0][0] * P[0].y;
#pragma unroll
            for (int e = 1; e < NQ; ++e) {
                X.x = fma(q[0][e], P[e].x, X.x);
                X.y = fma(q[0][e], P[e].y, X.y);
            }
#pragma unroll
            for (int a = 0; a <= FHH; ++a) {
                if (!((need >> a) & 1u)) {
                    if (a == FHH) ring[(PH + 2 * FHH) % FS] = make_double2(0.0, 0.0);
                    continue;
                }
                ring[(PH + FHH - a) % FS].x = fma(q[1][a], X.x, ring[(PH + FHH - a) % FS].x);
                ring[(PH + FHH - a) % FS].y = fma(q[1][a], X.y, ring[(PH + FHH - a) % FS].y);
                if (a == FHH) {
                    ring[(PH + 2 * FHH) % FS].x = q[1][a] * X.x;
                    ring[(PH + 2 * FHH) % FS].y = q[1][a] * X.y;
                } else if (a > 0) {
                    ring[(PH + FHH + a) % FS].x = fma(q[1][a], X.x, ring[(PH + FHH + a) % FS].x);
                    ring[(PH + FHH + a) % FS].y = fma(q[1][a], X.y, ring[(PH + FHH + a) % FS].y);
                }
            }
        } else {
#pragma unroll
        for (int a = 0; a <= FHH; ++a) {
            // tap rows FHH-a and FHH+a are equal: one dot product for the two output rows
            if (!((need >> a) & 1u)) {
                if (a == FHH) ring[(PH + 2 * FHH) % FS] = make_double2(0.0, 0.0);
                continue;
            }
            double2 T;
            T.x = q[a][0] * P[0].x;
            T.y = q[a][0] * P[0].y;
#pragma unroll
            for (int e = 1; e < NQ; ++e) {
                T.x = fma(q[a][e], P[e].x, T.x);
                T.y = fma(q[a][e], P[e].y, T.y);
            }
            ring[(PH + FHH - a) % FS].x += T.x;
            ring[(PH + FHH - a) % FS].y += T.y;
            if (a == FHH) {
                // the newest slot (the one that finished last step) starts here
                ring[(PH + 2 * FHH) % FS] = T;
            } else if (a > 0) {
                ring[(PH + FHH + a) % FS].x += T.x;
                ring[(PH + FHH + a) % FS].y += T.y;
            }
        }
        }
    }
    if (store) {
        double2 v = ring[PH % FS];
        if constexpr (LSF) {
            // LSF on the finished row: spectrum -> wave-private LDS buffer with a circular
            // halo of RL channels -> aligned 16-byte window reads
            double *ms = myspec + (size_t)(lane / HLS) * (DP + 2 * RL);   // this lane group's spectrum
            *reinterpret_cast<double2 *>(ms + RL + 2 * zl) = v;
            if (2 * zl < RL) *reinterpret_cast<double2 *>(ms + DP + RL + 2 * zl) = v;
            if (2 * zl >= DP - RL) *reinterpret_cast<double2 *>(ms + RL + 2 * zl - DP) = v;
            __builtin_amdgcn_wave_barrier();  // LDS is in order per wavefront
            const double *bt = ms + 2 * zl;
            double2 acc = make_double2(0.0, 0.0);
#pragma unroll
            for (int j = 0; j < RL + 1; ++j) {
                const double2 p = *reinterpret_cast<const double2 *>(bt + 2 * j);
                // p = (s[2j], s[2j+1]); acc.x = sum w[k] s[k], acc.y = sum w[k] s[k+1]
                if (2 * j <= 2 * RL) acc.x = fma(w[2 * j], p.x, acc.x);
                if (2 * j + 1 <= 2 * RL) acc.x = fma(w[2 * j + 1], p.y, acc.x);
                if (2 * j - 1 >= 0) acc.y = fma(w[2 * j - 1], p.x, acc.y);
                if (2 * j <= 2 * RL) acc.y = fma(w[2 * j], p.y, acc.y);
            }
            __builtin_amdgcn_wave_barrier();
            v = acc;
        }
        if constexpr (RESID) {
            v.x = dcur.x - v.x;
            v.y = dcur.y - v.y;
        }
        double2 *dst = reinterpret_cast<double2 *>(out + (long)oy * rowstride +
                                                   (long)x * (ZB ? cstride : DP) + 2 * zl);
        if (xok) *dst = v;
    }
}

// FS consecutive steps starting at step `base` (base mod FS == 0), a workgroup barrier
// before every even step: input rows travel in PAIRS (see the loader).
template <int FS, int NW, bool LSF, bool RESID, bool SEP, int DPS, bool ZB, int PH = 0>
__device__ __forceinline__ void conv_rows_steps(int base, int nsteps, int y0, int yend,
                                                long rowstride, int cstride, int x, bool xok, int wave,
                                                int lane,
                                                const double *rows, double *myspec,
                                                const double (&q)[(FS + 1) / 2][(FS + 1) / 2],
                                                const double (&w)[2 * LSF_RL + 1],
                                                const double *__restrict__ data,
                                                double *__restrict__ out, double2 (&ring)[FS],
                                                double2 &dnext) {
    if constexpr (PH < FS) {
        const int i = base + PH;
        if (i < nsteps) {
            // (every LDS read of the previous pair has been consumed: data dependences)
            if ((i & 1) == 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            conv_rows_step<FS, NW, LSF, RESID, SEP, PH, DPS, ZB>(i, y0, yend, rowstride, cstride, x, xok,
                                                                 wave, lane, rows, myspec, q, w, data,
                                                                 out, ring, dnext);
            // keep the steps apart: interleaving two of them costs more registers than the
            // 128 that four wavefronts per SIMD allow.  (Round 4: requesting the NEXT row's FS
            // values right after this row's fold -- the workgroup's wavefronts leave the
            // row-pair barrier together, wait on the LDS together, compute together: half their
            // resident cycles are parked on s_waitcnt, profiles/r04_issue_counters.txt -- needs
            // 44 more registers across the dot products: 128 VGPRs and 167-467 spills in every
            // instantiation.  Not kept.  Nor were the FHH + 1 dot products advancing together, tap
            // column by tap column (six independent chains, the centre column's products issued
            // while the outer columns are still on their way): 116 VGPRs, 62.8 against 54.4 us.)
            __builtin_amdgcn_sched_barrier(0);
        }
        conv_rows_steps<FS, NW, LSF, RESID, SEP, DPS, ZB, PH + 1>(base, nsteps, y0, yend, rowstride, cstride,
                                                                  x, xok, wave, lane, rows, myspec, q, w,
                                                                  data, out, ring, dnext);
    }
}

// TSYM = 2 selects the outer-product form (SEP): quad holds v[e] (row 0) and u[a] (row 1).
template <int FS, int NW, bool LSF, bool LSYM, bool RESID, int TSYM, int DPS = CONV_DP, bool ZB = false>
__global__ __launch_bounds__((NW + 1) * 64) void k_conv_rows(ConvRowsArgs A,
                                                              const double *__restrict__ in,
                                                              double *__restrict__ out,
                                                              const double *__restrict__ quad,
                                                              const double *__restrict__ wl,
                                                              const double *__restrict__ data) {
    static_assert(!ZB || (DPS == CONV_DP && !LSF), "z-blocks: 128-channel blocks, FSF only");
    constexpr int FHH = (FS - 1) / 2, NQ = FHH + 1, DP = DPS;
    constexpr int NBUF = CONV_NBUF, RL = LSF_RL;
    using Geo = ConvGeo<FS, NW, DPS>;
    constexpr int NCH = Geo::NCH, RBUF = Geo::RBUF, SPW = Geo::SPW, NWC = Geo::NWC, HLS = DPS / 2;
    extern __shared__ double smem[];
    double *rows = smem;                                // [NBUF][RBUF]: (NWC + FS - 1) spectra per row
    double *spec = rows + (size_t)NBUF * RBUF;          // [NW][SPW][DP + 2 RL]
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    // XCD-aware block order: consecutive LOGICAL blocks (x-neighbours of one row
    // strip, FS-1 common input columns) on one XCD and its L2 (bijective remap)
    int blk = blockIdx.x;
    if (A.xcd_remap) {
        const int nb = gridDim.x, qq = nb / 8, rm = nb % 8, xcd = blk % 8;
        blk = (xcd < rm ? xcd * (qq + 1) : rm * (qq + 1) + (xcd - rm) * qq) + blk / 8;
    }
    // ZB: z-block slowest -- the x-neighbours of one block and strip stay neighbours
    const int zb = ZB ? blk / (A.ngx * A.ngy) : 0;
    if (ZB) blk -= zb * (A.ngx * A.ngy);
    const int gy = blk / A.ngx, gx = blk - gy * A.ngx;
    const int x0 = gx * NWC, y0 = gy * A.HY;
    const int yend = min(y0 + A.HY, A.H);
    const int nsteps = (yend - y0) + 2 * FHH;
    const int npairs = (nsteps + 1) / 2;
    const int cstride = ZB ? A.Dp : DP;               // doubles between adjacent columns
    const long rowstride = (long)A.W * cstride;
    const bool zok = !ZB || zb * CONV_DP + 2 * lane < A.Dp;  // (Dp is even: a z-pair is in or out)
    if (ZB) {  // this block's channels
        in += zb * CONV_DP;
        out += zb * CONV_DP;
        if (RESID) data += zb * CONV_DP;
    }

    if (wave == NW) {
        // ---- loader wavefront: input rows travel in pairs ---------------------------
        // pair j = rows 2j, 2j+1 -> buffers (2j) % 4, (2j+1) % 4.  While the compute
        // waves read pair j, pair j+1 is in flight (<= 2 NC = 50 loads outstanding,
        // under the 63 a wavefront may have); its buffers held pair j-1, whose readers
        // passed the barrier that precedes the issue.
        // chunk c of a row segment: doubles [128 c, 128 c + 128) from column x0 - FHH on; lane l
        // holds doubles 128 c + 2 l, + 1, i.e. column xs + (128 c + 2 l) / DPS.  A lane whose
        // column lies outside the cube never loads: its LDS slot keeps the zero written here.
        const int xs = x0 - FHH;
        auto lane_col = [&](int c) { return xs + (c * CONV_DP + 2 * lane) / DP; };
        for (int c = 0; c < NCH; ++c) {
            const int xx = lane_col(c);
            if (xx < 0 || xx >= A.W || !zok)
                for (int b = 0; b < NBUF; ++b)
                    *reinterpret_cast<double2 *>(rows + (size_t)b * RBUF + (size_t)c * CONV_DP + 2 * lane) =
                        make_double2(0.0, 0.0);
        }
        auto issue = [&](int i) {
            const int r = y0 - FHH + i;
            if (i >= nsteps) return;
            double *dst = rows + (size_t)(i % NBUF) * RBUF;
            if (r < 0 || r >= A.H) {  // a row outside the cube is a row of zeros
                for (int c = 0; c < NCH; ++c)
                    *reinterpret_cast<double2 *>(dst + (size_t)c * CONV_DP + 2 * lane) =
                        make_double2(0.0, 0.0);
                return;
            }
            const double *src = in + (long)r * rowstride + (long)xs * cstride + 2 * lane;
#if defined(D3D_CONV_DIAG) && D3D_CONV_DIAG == 1
            return;  // diagnostic: no loads at all (compute floor)
#endif
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int xx = lane_col(c);
                // (ZB: chunk c is column xs + c, cstride doubles further on)
                if (xx >= 0 && xx < A.W && zok)
                    conv_glds16(src + (long)c * (ZB ? cstride : CONV_DP), dst + (size_t)c * CONV_DP);
            }
        };
        __builtin_amdgcn_s_setprio(3);  // the loads are on everybody's critical path (-1 us)
        issue(0);
        issue(1);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // zero slots written
        for (int j = 0; j < npairs; ++j) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // pair j has landed
            asm volatile("s_barrier" ::: "memory");           // readable; pair j-1's buffers free
            issue(2 * j + 2);
            issue(2 * j + 3);
        }
        return;
    }

    // ---- compute wavefront: output columns x0 + wave SPW + (lane group) -------------------
    const int x = x0 + wave * SPW + lane / HLS;
    const bool xok = x < A.W && zok;
    if (x0 + wave * SPW >= A.W) {  // columns past the cube's edge: only keep the barriers company
        asm volatile("s_barrier" ::: "memory");
        for (int j = 0; j < npairs; ++j) asm volatile("s_barrier" ::: "memory");
        return;
    }
    // taps and LSF weights: scalar registers for the whole march
    double q[NQ][NQ];
#pragma unroll
    for (int a = 0; a < NQ; ++a)
#pragma unroll
        for (int m = 0; m < NQ; ++m)
            q[a][m] = ((TSYM == 1 && m > a) || (TSYM == 2 && a > 1)) ? 0.0 : quad[a * NQ + m];
    if constexpr (TSYM == 1) {
#pragma unroll
        for (int a = 0; a < NQ; ++a)
#pragma unroll
            for (int m = a + 1; m < NQ; ++m) q[a][m] = q[m][a];
    }
    double w[2 * RL + 1];
#pragma unroll
    for (int j = 0; j <= 2 * RL; ++j) w[j] = 0.0;
    if constexpr (LSF) {
#pragma unroll
        for (int j = 0; j <= 2 * RL; ++j) w[j] = (LSYM && j > RL) ? 0.0 : wl[j];
        if constexpr (LSYM) {
#pragma unroll
            for (int j = RL + 1; j <= 2 * RL; ++j) w[j] = w[2 * RL - j];
        }
    }
    double2 ring[FS];
#pragma unroll
    for (int k = 0; k < FS; ++k) ring[k] = make_double2(0.0, 0.0);
    double *myspec = spec + (size_t)wave * Geo::SPEC;
    double2 dnext = make_double2(0.0, 0.0);  // RESID: data of the row finished next
    asm volatile("s_barrier" ::: "memory");  // prologue
    // blocks of FS steps (the ring rotates through static registers)
    for (int base = 0; base < nsteps; base += FS)
        conv_rows_steps<FS, NW, LSF, RESID, TSYM == 2, DPS, ZB>(base, nsteps, y0, yend, rowstride, cstride, x,
                                                                xok, wave, lane, rows, myspec, q, w, data,
                                                                out, ring, dnext);
}

}  // namespace d3d
