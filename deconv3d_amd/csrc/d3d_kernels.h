// HIP kernels of the deconv3d likelihood path for gfx950 (MI355X, wave64).
//
// Device layout.  The reference stores cubes (D,H,W) with x fastest
// (lib/run.py:146-149).  On the device every cube is stored SPECTRUM-CONTIGUOUS,
// (H, W, Dp) with Dp = D rounded up to even and zero padding: one spaxel's
// spectrum is one contiguous run (1 KiB at D=128), so
//   * the MH update's FSF window is fh runs of fw*Dp contiguous doubles,
//   * the spectral (LSF) pass has z along the lanes of a wavefront,
//   * the spatial (FSF) pass reads whole spectra, 16 B per lane, coalesced.
// A thread always owns a z-PAIR (one double2 = one 16-byte access).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "d3d_rng.h"

namespace d3d {

// ------------------------------------------------------------------------- //
// layout conversion                                                          //
// ------------------------------------------------------------------------- //

// (D, HW) host layout -> (HW, Dp) device layout, zero padded.  32x32 LDS tile.
static __global__ __launch_bounds__(256) void k_to_device_layout(const double *__restrict__ src,
                                                           double *__restrict__ dst, int D,
                                                           int Dp, long HW) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const long s0 = (long)blockIdx.x * 32;
    const int z0 = blockIdx.y * 32;
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
        const int z = z0 + ty + k;
        const long s = s0 + tx;
        tile[ty + k][tx] = (z < D && s < HW) ? src[(long)z * HW + s] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
        const long s = s0 + ty + k;
        const int z = z0 + tx;
        if (s < HW && z < Dp) dst[s * Dp + z] = tile[tx][ty + k];
    }
}

// (HW, Dp) device layout -> (D, HW) host layout.
static __global__ __launch_bounds__(256) void k_to_host_layout(const double *__restrict__ src,
                                                         double *__restrict__ dst, int D, int Dp,
                                                         long HW) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long s0 = (long)blockIdx.x * 32;
    const int z0 = blockIdx.y * 32;
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
        const long s = s0 + ty + k;
        const int z = z0 + tx;
        tile[ty + k][tx] = (s < HW && z < Dp) ? src[s * Dp + z] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 32; k += 8) {
        const int z = z0 + ty + k;
        const long s = s0 + tx;
        if (z < D && s < HW) dst[(long)z * HW + s] = tile[tx][ty + k];
    }
}

// data/variance fix-ups of lib/run.py:171-200 + SURVEY appendix A, elementwise
// in host layout: var==0 -> 1e12; NaN voxel -> data 0, 1/var 0.
static __global__ void k_prepare_data(double *__restrict__ data, double *__restrict__ ivar,
                               const double *__restrict__ var, double var_scalar, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = var ? var[i] : var_scalar;
    double d = data[i];
    if (v == 0.0) v = 1e12;
    double iv = 1.0 / v;
    if (d != d || v != v) {
        d = 0.0;
        iv = 0.0;
    }
    data[i] = d;
    ivar[i] = iv;
}

// ------------------------------------------------------------------------- //
// line model + spectral (LSF) pass                                           //
// ------------------------------------------------------------------------- //

// lib/line_models.py:109 with unit amplitude; w == 0 would be 0/0 in the
// reference -- here it degenerates to a delta at z == c.
__device__ __forceinline__ double unit_gaussian(double z, double c, double w) {
    const double w2 = 2.0 * w * w;
    const double d = z - c;
    return (w2 > 0.0) ? exp(-(d * d) / w2) : ((d == 0.0) ? 1.0 : 0.0);
}

struct SpectralArgs {
    int D, Dp, HL, N, ntaps;
    long nspax;
    const int *shift;      // [ntaps]  (N/2 - h - t) mod N
    const double *weight;  // [ntaps]
};

// Closed form of convolve_1d (lib/convolution.py:89-120):
// out[k] = sum_t w_t * ext[(k + s_t) mod N], ext zero beyond D, N power of 2.
__device__ __forceinline__ double lsf_apply(const double *ext, int k, const SpectralArgs &A) {
    double acc = 0.0;
    for (int t = 0; t < A.ntaps; ++t) acc += A.weight[t] * ext[(k + A.shift[t]) & (A.N - 1)];
    return acc;
}

// params (H,W,3) -> cube of LSF-convolved lines (mode 1) or clean lines
// (mode 0), zero where mask == 0.  lib/run.py:597-621 and :1011-1024.
// One group of HL threads per spaxel, G groups per block, ext[] in LDS.
template <int NT>
__global__ __launch_bounds__(NT) void k_lines(SpectralArgs A, const double *__restrict__ params,
                                              const uint8_t *__restrict__ mask,
                                              double *__restrict__ out, int convolved) {
    extern __shared__ double smem[];
    const int G = NT / A.HL;
    const int g = threadIdx.x / A.HL, zl = threadIdx.x - g * A.HL;
    const long sp = (long)blockIdx.x * G + g;
    const bool active = (g < G) && (sp < A.nspax);
    double *ext = smem + (size_t)g * A.N;
    double a = 0, c = 0, w = 1;
    bool live = false;
    if (active) {
        live = mask[sp] != 0;
        a = params[sp * 3 + 0];
        c = params[sp * 3 + 1];
        w = params[sp * 3 + 2];
    }
    const bool use_lsf = convolved && A.ntaps > 0;
    double2 v = make_double2(0.0, 0.0);
    if (active && live) {
        const int z = 2 * zl;
        v.x = (z < A.D) ? a * unit_gaussian((double)z, c, w) : 0.0;
        v.y = (z + 1 < A.D) ? a * unit_gaussian((double)(z + 1), c, w) : 0.0;
    }
    if (use_lsf) {
        if (active) {
            for (int j = 2 * zl; j < A.N; j += 2 * A.HL) {
                ext[j] = 0.0;
                ext[j + 1] = 0.0;
            }
        }
        __syncthreads();
        if (active) {
            ext[2 * zl] = v.x;
            ext[2 * zl + 1] = v.y;
        }
        __syncthreads();
        if (active && live) {
            const int z = 2 * zl;
            v.x = (z < A.D) ? lsf_apply(ext, z, A) : 0.0;
            v.y = (z + 1 < A.D) ? lsf_apply(ext, z + 1, A) : 0.0;
        }
    }
    if (active) *reinterpret_cast<double2 *>(out + sp * A.Dp + 2 * zl) = v;
}

// Spectral pass of an arbitrary cube: out[sp,:] = LSF (*) in[sp,:].
template <int NT>
__global__ __launch_bounds__(NT) void k_spectral(SpectralArgs A, const double *__restrict__ in,
                                                 double *__restrict__ out) {
    extern __shared__ double smem[];
    const int G = NT / A.HL;
    const int g = threadIdx.x / A.HL, zl = threadIdx.x - g * A.HL;
    const long sp = (long)blockIdx.x * G + g;
    const bool active = (g < G) && (sp < A.nspax);
    double *ext = smem + (size_t)g * A.N;
    double2 v = make_double2(0.0, 0.0);
    if (active) {
        v = *reinterpret_cast<const double2 *>(in + sp * A.Dp + 2 * zl);
        for (int j = 2 * zl; j < A.N; j += 2 * A.HL) {
            ext[j] = 0.0;
            ext[j + 1] = 0.0;
        }
    }
    __syncthreads();
    if (active) {
        const int z = 2 * zl;
        ext[z] = (z < A.D) ? v.x : 0.0;
        ext[z + 1] = (z + 1 < A.D) ? v.y : 0.0;
    }
    __syncthreads();
    if (active) {
        const int z = 2 * zl;
        v.x = (z < A.D) ? lsf_apply(ext, z, A) : 0.0;
        v.y = (z + 1 < A.D) ? lsf_apply(ext, z + 1, A) : 0.0;
        *reinterpret_cast<double2 *>(out + sp * A.Dp + 2 * zl) = v;
    }
}

constexpr int LSF_RL = 8;  // radius of the LSF taps the dense / fused forms handle

// Fast spectral pass for power-of-two depths whose LSF taps lie within +-LSF_RL
// channels and whose spectrum fits one wavefront (Dp <= 128):
//   out[k] = sum_j wl[j] * v[(k + j - LSF_RL) mod Dp]
// (closed form of convolve_1d, lib/convolution.py:89-120).  The spectrum goes to
// a wave-private LDS buffer with a circular halo, every lane reads back its
// aligned window of 2*LSF_RL+2 channels with 16-byte reads (no bank conflicts,
// no block barrier) and applies the dense taps.  Pure streaming: HBM-bound.
template <int NT>
__global__ __launch_bounds__(NT) void k_spectral_dense(int Dp, int HL, long nspax,
                                                       const double *__restrict__ wl,
                                                       const double *__restrict__ in,
                                                       double *__restrict__ out) {
    extern __shared__ double smem[];
    constexpr int RL = LSF_RL;
    const int G = NT / HL;
    const int g = threadIdx.x / HL, zl = threadIdx.x - g * HL;
    const long sp = (long)blockIdx.x * G + g;
    if (g >= G || sp >= nspax) return;
    double *buf = smem + (size_t)g * (Dp + 2 * RL);
    const double2 v = *reinterpret_cast<const double2 *>(in + sp * Dp + 2 * zl);
    *reinterpret_cast<double2 *>(buf + RL + 2 * zl) = v;
    if (2 * zl < RL) *reinterpret_cast<double2 *>(buf + Dp + RL + 2 * zl) = v;
    if (2 * zl >= Dp - RL) *reinterpret_cast<double2 *>(buf + RL + 2 * zl - Dp) = v;
    __builtin_amdgcn_wave_barrier();  // wave-private buffer: LDS is in order per wave
    double w[2 * RL + 2];
#pragma unroll
    for (int j = 0; j < RL + 1; ++j) {
        const double2 p = *reinterpret_cast<const double2 *>(buf + 2 * zl + 2 * j);
        w[2 * j] = p.x;
        w[2 * j + 1] = p.y;
    }
    double2 acc = make_double2(0.0, 0.0);
#pragma unroll
    for (int j = 0; j < 2 * RL + 1; ++j) {
        const double t = wl[j];
        acc.x = fma(t, w[j], acc.x);
        acc.y = fma(t, w[j + 1], acc.y);
    }
    *reinterpret_cast<double2 *>(out + sp * Dp + 2 * zl) = acc;
}

// k_lines with the dense LSF (round 4): a group of HLG lanes of ONE wavefront (64 above 64
// channels, else the power of two that holds the spectrum: 64 / HLG spaxels per wavefront)
// builds its spaxel's line -- 2 HLG channels per step, `steps` steps -- into a wave-private LDS
// buffer  ext[-LSF_RL, span + LSF_RL),  span = steps 2 HLG <= N:  the zero-extended,
// N-periodic spectrum of convolve_1d's closed form (lib/convolution.py:89-160), whose two halos
// are copies of line channels (the wrap of the padded grid) or zero.  Then every lane reads its
// aligned window of 2 LSF_RL + 2 channels with 16-byte reads and applies the dense taps IN THE
// ORDER OF THE TAP LIST (descending shift), so the cube equals k_lines' bit for bit.  No block
// barrier, no index arithmetic per tap: 235 instead of 430 instructions per channel pair
// (300x300x256: 114 -> 76 us).
//
// FAST (option lines_dense = 2, NOT the default): the line through a reciprocal and an own exp
// for arguments <= 0 (13th-degree Taylor polynomial on |r| <= ln2 / 2 after Cody-Waite
// reduction, ~ 45 instead of ~ 65 instructions per channel): within 2 ulp of unit_gaussian's
// correctly divided, library exp.  Measured 7-10 % faster (300x300x128 without the LSF 33.5
// against 35.9 us, x256 with it 68 against 76): the kernel is bound by the dependent fp64
// chain of the exp, not by its instruction count (without its stores 27.7 against 30.3 us;
// without the exp, stores only, 23; neither 13) -- not worth a line cube that differs from
// the sweep kernels' unit_gaussian in the last bits.
__device__ __forceinline__ double exp_nonpositive(double x) {
    x = x < -800.0 ? -800.0 : x;  // (e^-800 is zero in fp64; NaN stays NaN)
    const double k = rint(x * 1.44269504088896338700e+00);
    double r = fma(k, -6.93147180369123816490e-01, x);
    r = fma(k, -1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;   // 1 / 13!
    p = fma(p, r, 2.08767569878681e-09);     // 1 / 12!
    p = fma(p, r, 2.505210838544172e-08);    // 1 / 11!
    p = fma(p, r, 2.755731922398589e-07);    // 1 / 10!
    p = fma(p, r, 2.7557319223985893e-06);   // 1 / 9!
    p = fma(p, r, 2.48015873015873e-05);     // 1 / 8!
    p = fma(p, r, 1.984126984126984e-04);    // 1 / 7!
    p = fma(p, r, 1.388888888888889e-03);    // 1 / 6!
    p = fma(p, r, 8.333333333333333e-03);    // 1 / 5!
    p = fma(p, r, 4.1666666666666664e-02);   // 1 / 4!
    p = fma(p, r, 1.6666666666666666e-01);   // 1 / 3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)k);
}

template <bool FAST>
static __global__ __launch_bounds__(256) void k_lines_dense(SpectralArgs A, int HLG, int steps, int L,
                                                            const double *__restrict__ wl,
                                                            const double *__restrict__ params,
                                                            const uint8_t *__restrict__ mask,
                                                            double *__restrict__ out, int convolved) {
    extern __shared__ double smem[];
    constexpr int RL = LSF_RL;
    const int S = 64 / HLG;  // spaxels per wavefront and round
    const int lane = threadIdx.x & 63;
    const int g = lane / HLG, zl = lane - g * HLG;
    // a wavefront takes L rounds of S consecutive spaxels, S L <= 64 (a wavefront per spaxel
    // lives for one memory round trip and ~ 150 instructions: that launch is bound by the
    // latency of its parameter loads at the occupancy it has).  Lane j loads the parameters of
    // spaxel j of the wavefront ONCE; the rounds take them by shuffle, so nothing waits on
    // memory between the stores of one round and the channels of the next.
    const long spbase = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * S * L;
    const int span = steps * 2 * HLG;
    double *buf = smem + (size_t)((threadIdx.x >> 6) * S + g) * (span + 2 * RL) + RL;  // -> channel 0
    const bool use_lsf = convolved && A.ntaps > 0;
    double t[2 * RL + 1];
    if (use_lsf) {
#pragma unroll
        for (int j = 0; j < 2 * RL + 1; ++j) t[j] = wl[j];
    }
    double pa = 0, pc = 0, pw = 1;
    int plive = 0;
    if (lane < S * L && spbase + lane < A.nspax) {
        plive = mask[spbase + lane] != 0;
        pa = params[(spbase + lane) * 3 + 0];
        pc = params[(spbase + lane) * 3 + 1];
        pw = params[(spbase + lane) * 3 + 2];
    }
    for (int it = 0; it < L; ++it) {
        if (spbase + (long)it * S >= A.nspax) break;  // (wave-uniform)
        const int src = it * S + g;
        const long sp = spbase + src;
        const bool active = sp < A.nspax;
        const double a = __shfl(pa, src), c = __shfl(pc, src), w = __shfl(pw, src);
        const bool live = __shfl(plive, src) != 0;
        const double inv_w2 = 1.0 / (2.0 * w * w);  // (FAST; w == 0 or a denormal 2 w^2: the delta at z == c)
        auto line_at = [&](int zi) {
            if (!FAST) return a * unit_gaussian((double)zi, c, w);
            const double d = (double)zi - c;
            return inv_w2 <= 1.79769313486231570815e+308 ? a * exp_nonpositive(-(d * d) * inv_w2)
                                                         : (d == 0.0 ? a : a * 0.0);
        };
        for (int b = 0; b < steps; ++b) {
            const int z = b * 2 * HLG + 2 * zl;
            double2 v = make_double2(0.0, 0.0);
            if (live) {
                v.x = (z < A.D) ? line_at(z) : 0.0;
                v.y = (z + 1 < A.D) ? line_at(z + 1) : 0.0;
            }
            if (use_lsf) *reinterpret_cast<double2 *>(buf + z) = v;
            else if (active && z < A.Dp) *reinterpret_cast<double2 *>(out + sp * A.Dp + z) = v;
        }
        if (!use_lsf) continue;
        __builtin_amdgcn_wave_barrier();  // wave-private buffer: LDS is in order per wave
        if (zl < RL) {  // the halos: pairs at m = -RL + 2 zl (front), span + 2 (zl - RL / 2) (back)
            const int m = zl < RL / 2 ? -RL + 2 * zl : span + 2 * (zl - RL / 2);
            int idx = m % A.N;
            if (idx < 0) idx += A.N;
            // (idx even: idx < D leaves idx + 1 <= span - 1, a channel the steps above wrote)
            const double2 h = idx < A.D ? *reinterpret_cast<const double2 *>(buf + idx) : make_double2(0.0, 0.0);
            *reinterpret_cast<double2 *>(buf + m) = h;
        }
        __builtin_amdgcn_wave_barrier();
        for (int b = 0; b < steps; ++b) {
            const int z = b * 2 * HLG + 2 * zl;
            if (z >= A.Dp) break;
            double win[2 * RL + 2];
#pragma unroll
            for (int j = 0; j < RL + 1; ++j) {
                const double2 p = *reinterpret_cast<const double2 *>(buf + z - RL + 2 * j);
                win[2 * j] = p.x;
                win[2 * j + 1] = p.y;
            }
            double2 acc = make_double2(0.0, 0.0);
#pragma unroll
            for (int j = 2 * RL; j >= 0; --j) {
                acc.x = fma(t[j], win[j], acc.x);
                acc.y = fma(t[j], win[j + 1], acc.y);
            }
            if (!live || z >= A.D) acc.x = 0.0;
            if (!live || z + 1 >= A.D) acc.y = 0.0;
            if (active) *reinterpret_cast<double2 *>(out + sp * A.Dp + z) = acc;
        }
        __builtin_amdgcn_wave_barrier();  // (the next round rewrites the buffer)
    }
}

// The dense form for ANY depth (round 3): a wavefront takes one 128-channel BLOCK of one
// spectrum, [z0, z0 + 128), and the LSF_RL channels either side of it,
//   out[k] = sum_j wl[j] * ext[(k + j - LSF_RL) mod N],  ext = the spectrum zero-extended to N
// (closed form of convolve_1d for every depth: lib/convolution.py:89-160 -- the wrap of the
// power-of-two padded grid, including the partial wrap of depths within LSF_RL of N, is the
// "mod N, zero beyond D" of the halo loads).  Wave-private LDS window, no block barrier, no
// limit on the depth: streaming, HBM-bound.  The halo channels are read a second time (from
// L2: the neighbouring block's wavefront reads them too): 144 channels per 128.
static __global__ __launch_bounds__(256) void k_spectral_blocks(int D, int Dp, int N, int nzb, long nwaves,
                                                                const double *__restrict__ wl,
                                                                const double *__restrict__ in,
                                                                double *__restrict__ out) {
    __shared__ double smem[4 * (128 + 2 * LSF_RL)];
    constexpr int RL = LSF_RL;
    const int lane = threadIdx.x & 63;
    const long wv = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wv >= nwaves) return;
    const long sp = wv / nzb;
    const int z0 = (int)(wv - sp * nzb) * 128;
    const double *src = in + sp * Dp;
    double *buf = smem + (size_t)(threadIdx.x >> 6) * (128 + 2 * RL);
    // value pair at (even) position m of the zero-extended, N-periodic spectrum
    auto fetch = [&](int m) {
        int idx = m % N;
        if (idx < 0) idx += N;
        double2 v = make_double2(0.0, 0.0);
        if (idx < D) {  // (idx even, Dp even: idx + 1 < Dp; the padding channel holds zero)
            v = *reinterpret_cast<const double2 *>(src + idx);
            if (idx + 1 >= D) v.y = 0.0;
        }
        return v;
    };
    *reinterpret_cast<double2 *>(buf + RL + 2 * lane) = fetch(z0 + 2 * lane);
    if (lane < RL / 2) *reinterpret_cast<double2 *>(buf + 2 * lane) = fetch(z0 - RL + 2 * lane);
    else if (lane < RL) *reinterpret_cast<double2 *>(buf + 128 + RL + 2 * (lane - RL / 2)) =
        fetch(z0 + 128 + 2 * (lane - RL / 2));
    __builtin_amdgcn_wave_barrier();  // wave-private buffer: LDS is in order per wave
    double w[2 * RL + 2];
#pragma unroll
    for (int j = 0; j < RL + 1; ++j) {
        const double2 p = *reinterpret_cast<const double2 *>(buf + 2 * lane + 2 * j);
        w[2 * j] = p.x;
        w[2 * j + 1] = p.y;
    }
    double2 acc = make_double2(0.0, 0.0);
#pragma unroll
    for (int j = 0; j < 2 * RL + 1; ++j) {
        const double t = wl[j];
        acc.x = fma(t, w[j], acc.x);
        acc.y = fma(t, w[j + 1], acc.y);
    }
    const int z = z0 + 2 * lane;
    if (z < Dp) {
        if (z >= D) acc.x = 0.0;  // the padding channel of an odd depth stays zero
        if (z + 1 >= D) acc.y = 0.0;
        *reinterpret_cast<double2 *>(out + sp * Dp + z) = acc;
    }
}

// The same pass with wavefront shuffles instead of the LDS window (HL == 64: one
// spectrum per wavefront, so the circular wrap is the wrap of the lane index):
// lane l needs channels 2l-8 .. 2l+9, i.e. both components of lanes l-4 .. l+4,
// fetched with ds_bpermute (__shfl).  Measured against the LDS form on MI355X
// (D3D_SPECTRAL_SHFL=1): both run at the streaming rate of the part; the pass is
// bound by HBM, not by how the neighbours are exchanged.
template <int NT>
__global__ __launch_bounds__(NT) void k_spectral_shfl(int Dp, long nspax,
                                                      const double *__restrict__ wl,
                                                      const double *__restrict__ in,
                                                      double *__restrict__ out) {
    constexpr int RL = LSF_RL;
    constexpr int G = NT / 64;
    const int g = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long sp = (long)blockIdx.x * G + g;
    if (sp >= nspax) return;
    const double2 v = *reinterpret_cast<const double2 *>(in + sp * Dp + 2 * lane);
    // w[j] = channel 2*lane - RL + j, j = 0 .. 2*RL + 1
    double w[2 * RL + 2];
#pragma unroll
    for (int m = -RL / 2; m <= RL / 2; ++m) {
        const int src = (lane + m) & 63;
        w[RL + 2 * m] = (m == 0) ? v.x : __shfl(v.x, src);
        w[RL + 2 * m + 1] = (m == 0) ? v.y : __shfl(v.y, src);
    }
    double2 acc = make_double2(0.0, 0.0);
#pragma unroll
    for (int j = 0; j < 2 * RL + 1; ++j) {
        const double t = wl[j];
        acc.x = fma(t, w[j], acc.x);
        acc.y = fma(t, w[j + 1], acc.y);
    }
    *reinterpret_cast<double2 *>(out + sp * Dp + 2 * lane) = acc;
}

// ------------------------------------------------------------------------- //
// spatial (FSF) pass: true 2-D convolution, zero boundary, 'same' size        //
// (scipy.signal.convolve2d(..., mode='same') of lib/run.py:1027-1029)         //
//   out[Y,X] = sum_{j,i} fsf[j,i] * in[Y - j + fhh, X - i + fhw]              //
// ------------------------------------------------------------------------- //

struct SpatialArgs {
    int Dp, HL, H, W, fh, fw;
    const double *fsf;   // [fh*fw]
    const double *data;  // residual epilogue: out = data - conv (or NULL)
    // fused spectral epilogue of the march kernel (or NULL): dense LSF weights
    // wl[j], j = 0..2*LSF_RL, so that out[k] = sum_j wl[j] * v[(k + j - LSF_RL) mod Dp]
    // (closed form of convolve_1d for power-of-two depths, lib/convolution.py:89-120)
    const double *lsf_dense;
    const double *sep_uv;  // k_spatial_sep: u[fh] | v[fw] with fsf == u v^T (or NULL)
    int xcd_remap;  // XCD-aware block order in the march kernel
    int alt_dir;    // alternate the march direction of vertically adjacent strips
    int stagger;    // start delay (x ~2000 cycles) of every other workgroup, 0 = none
    // diagnostic build only (D3D_STAMP=1): per-wavefront cycle sums of the
    // phases of a march step; a buffer of its own, never read by product code
    unsigned long long *dbg;
};

// Register-tiled: a thread owns one z-pair and TX consecutive x outputs of one
// row; per tap row it loads TX+FW-1 inputs and issues FW*TX double2 FMAs.
// Taps are wave-uniform (scalar loads).  Lanes run along z: every global
// access is a contiguous 16 B/lane run.
template <int NT, int FW, int TX>
__global__ __launch_bounds__(NT) void k_spatial(SpatialArgs A, const double *__restrict__ in,
                                                double *__restrict__ out) {
    const int S = NT / A.HL;  // strips per block
    const int s = threadIdx.x / A.HL, zl = threadIdx.x - s * A.HL;
    const int nxs = (A.W + TX - 1) / TX;
    const long strip = (long)blockIdx.x * S + s;
    if (s >= S || strip >= (long)A.H * nxs) return;
    const int y = (int)(strip / nxs);
    const int x0 = (int)(strip - (long)y * nxs) * TX;
    constexpr int FHW = (FW - 1) / 2;
    const int fhh = (A.fh - 1) / 2;

    double2 acc[TX];
#pragma unroll
    for (int t = 0; t < TX; ++t) acc[t] = make_double2(0.0, 0.0);

    for (int j = 0; j < A.fh; ++j) {
        const int yy = y - j + fhh;
        if (yy < 0 || yy >= A.H) continue;
        double2 row[TX + FW - 1];
        const double *base = in + ((long)yy * A.W) * A.Dp + 2 * zl;
#pragma unroll
        for (int i = 0; i < TX + FW - 1; ++i) {
            const int xx = x0 - FHW + i;
            row[i] = (xx >= 0 && xx < A.W) ? *reinterpret_cast<const double2 *>(base + (long)xx * A.Dp)
                                           : make_double2(0.0, 0.0);
        }
        const double *taps = A.fsf + j * FW;
#pragma unroll
        for (int i = 0; i < FW; ++i) {
            const double tap = taps[i];
#pragma unroll
            for (int t = 0; t < TX; ++t) {
                acc[t].x = fma(tap, row[t + FW - 1 - i].x, acc[t].x);
                acc[t].y = fma(tap, row[t + FW - 1 - i].y, acc[t].y);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < TX; ++t) {
        const int xo = x0 + t;
        if (xo < A.W) {
            const long o = ((long)y * A.W + xo) * A.Dp + 2 * zl;
            double2 r = acc[t];
            if (A.data) {
                const double2 d = *reinterpret_cast<const double2 *>(A.data + o);
                r.x = d.x - r.x;
                r.y = d.y - r.y;
            }
            *reinterpret_cast<double2 *>(out + o) = r;
        }
    }
}

// Column march (the fast path for square FSFs).  A thread owns one z-pair and
// TX output columns and walks down HY output rows.  It keeps the FS output rows
// that the current input row touches as a register ring: input row r feeds ring
// slot k (output row r - FHH + k) with tap row k, slot 0 is complete after the
// step, is stored, and the ring shifts down one slot.  Every input row is
// loaded ONCE per strip (TX+FS-1 loads for FS*FS*TX double2 FMAs); the taps are
// staged in LDS and read as broadcasts; all global accesses are 16 B/lane
// contiguous along z.
//   UNI : HL is a multiple of 64, i.e. a wavefront works on ONE strip: strip
//         coordinates are made scalar (readfirstlane) so that every range test
//         is a scalar branch and the steady state (all FS slots live, strip
//         interior in x) is one branch-free block.
//   SYMX: the FSF is mirror-symmetric in x (fsf[k][i] == fsf[k][FS-1-i], true
//         for every Gaussian/Moffat with pa = 0): the mirrored inputs are
//         summed once per input row and shared by all FS slots, (FS+1)/2 FMAs
//         per tap row instead of FS.
__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

template <int NT, int FS, int TX, bool SYMX, bool UNI, bool FUSE, bool SYMY, bool STAMP = false>
__global__ __launch_bounds__(NT, (NT <= 512 ? 2 : 4)) void k_spatial_march(
    SpatialArgs A, const double *__restrict__ in, double *__restrict__ out, int HY) {
    constexpr int FHH = (FS - 1) / 2;
    constexpr int NP = SYMX ? (FHH + 1) : FS;  // products per (slot, column)
    constexpr int NR = TX + FS - 1;            // inputs per row
    __shared__ double s_taps[FS * FS + FS];    // + one pad row for the prefetch
    // fused LSF epilogue: one private spectrum buffer (with circular halo) per
    // strip and output column; a strip never straddles wavefronts when FUSE
    extern __shared__ double s_spec[];
    for (int i = threadIdx.x; i < FS * FS + FS; i += NT) s_taps[i] = i < FS * FS ? A.fsf[i] : 0.0;
    __syncthreads();

    const int S = NT / A.HL;
    int s = threadIdx.x / A.HL;
    const int zl = threadIdx.x - s * A.HL;
    if constexpr (UNI) s = __builtin_amdgcn_readfirstlane(s);
    const int nxs = (A.W + TX - 1) / TX;
    const int nys = (A.H + HY - 1) / HY;
    // XCD-aware block order: workgroups b and b+8 share an XCD (and its L2), so
    // consecutive LOGICAL blocks -- x-neighbours of one row strip, which read
    // the same input rows at the same time with FS-1 common columns -- are laid
    // on one XCD (bijective remap, cdna guide T1).
    int blk = blockIdx.x;
    if (A.xcd_remap) {
        const int nb = gridDim.x, q = nb / 8, rm = nb % 8, xcd = blk % 8;
        blk = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + blk / 8;
    }
    const long item = (long)blk * S + s;
    if (s >= S || item >= (long)nxs * nys) return;
    const int ys = (int)(item / nxs);
    const int x0 = (int)(item - (long)ys * nxs) * TX;
    const int y0 = ys * HY;
    const int yend = min(y0 + HY, A.H);
    const long rowstride = (long)A.W * A.Dp;
    const bool xin = (x0 - FHH >= 0) && (x0 + TX - 1 + FHH < A.W);

    double2 ring[FS][TX];
#pragma unroll
    for (int k = 0; k < FS; ++k)
#pragma unroll
        for (int t = 0; t < TX; ++t) ring[k][t] = make_double2(0.0, 0.0);

    // March direction: with a y-symmetric FSF the stencil is invariant under
    // y -> -y, so odd row strips walk UP.  Vertically adjacent strips then read
    // their FS-1 shared halo rows at the same time (both at their start or both
    // at their end) and, being on the same XCD (remap above), share them in L2:
    // measured fetch traffic 265 MB -> 131 MB per launch (L2 hit 32 % -> 56 %).
    const int dir = (SYMY && A.alt_dir && (ys & 1)) ? -1 : 1;
    const int nsteps = (yend - y0) + 2 * FHH;
    int r = dir > 0 ? y0 - FHH : yend - 1 + FHH;
    // ROT (x+y symmetric path): the march is unrolled by FS and ring slot k of
    // phase ph lives in physical register (ph + k) % FS -- the slot of a given
    // output row never moves, so the FS-1 ring moves per step disappear (16 %
    // of the kernel's fp64-rate instructions).
    constexpr bool ROT = SYMX && SYMY && !FUSE;
    constexpr int UNR = ROT ? FS : 1;
    // Stagger: co-resident wavefronts run the same program and fall into lock
    // step (all issue their row loads together, then all compute together).
    // Delaying every other workgroup by about half a step lets one wave's loads
    // overlap its SIMD partner's FMAs (cdna guide, 'try a stagger').
    if (A.stagger > 0 && (((blk >> 3) ^ blk) & 1)) {
        for (int i = 0; i < A.stagger; ++i) __builtin_amdgcn_s_sleep(32);
    }
    unsigned long long acc_issue = 0, acc_wait = 0, acc_math = 0, acc_tail = 0, t0 = 0, t1 = 0,
                       t2 = 0, t3 = 0;
    const unsigned long long t_begin = STAMP ? stamp() : 0;
    for (int sbase = 0; sbase < nsteps; sbase += UNR)
#pragma unroll
    for (int ph = 0; ph < UNR; ++ph) {
        const int step = sbase + ph;
        if (step >= nsteps) continue;
        if constexpr (STAMP) t0 = stamp();
        if (r >= 0 && r < A.H) {
            const double *base = in + (long)r * rowstride + (long)(x0 - FHH) * A.Dp + 2 * zl;
            double2 row[NR];
            if (xin) {
#pragma unroll
                for (int i = 0; i < NR; ++i)
                    row[i] = *reinterpret_cast<const double2 *>(base + (long)i * A.Dp);
            } else {
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    const int xx = x0 - FHH + i;
                    row[i] = (xx >= 0 && xx < A.W)
                                 ? *reinterpret_cast<const double2 *>(base + (long)i * A.Dp)
                                 : make_double2(0.0, 0.0);
                }
            }
            if constexpr (STAMP) {
                t1 = stamp();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                t2 = stamp();
                acc_issue += t1 - t0;
                acc_wait += t2 - t1;
            }
            // Taps are re-read from LDS every step (an opaque zero keeps the
            // compiler from hoisting FS*FS loop-invariant registers out of the
            // row loop), one tap row ahead of the FMAs that use it.
            int opq = 0;
            asm volatile("" : "+v"(opq));
            const double *taps = s_taps + opq;
            double tcur[NP], tnxt[NP];
#pragma unroll
            for (int m = 0; m < NP; ++m) tcur[m] = taps[m];
            // SYMX: P[t][m] = row[t+m] + row[t+FS-1-m] (m < FHH), centre row[t+FHH]
            double2 P[SYMX ? TX : 1][SYMX ? NP : 1];
            if constexpr (SYMX) {
#pragma unroll
                for (int t = 0; t < TX; ++t) {
#pragma unroll
                    for (int m = 0; m < FHH; ++m) {
                        P[t][m].x = row[t + m].x + row[t + FS - 1 - m].x;
                        P[t][m].y = row[t + m].y + row[t + FS - 1 - m].y;
                    }
                    P[t][FHH] = row[t + FHH];
                }
            }
            if constexpr (SYMX && SYMY) {
                // FSF also mirror-symmetric in y (fsf[k][i] == fsf[FS-1-k][i]): tap
                // rows FHH-a and FHH+a are the same, so their dot product with the
                // folded inputs, T_a = sum_m fsf[FHH-a][m] * P[.][m], is formed once
                // and ADDED to the two ring slots FHH-a (output row r-a) and FHH+a
                // (output row r+a): (FHH+1)^2 FMAs + FS adds per column instead of
                // FS*(FHH+1) FMAs.
#pragma unroll
                for (int m = 0; m < NP; ++m) tcur[m] = taps[FHH * FS + m];
#pragma unroll
                for (int a = 0; a <= FHH; ++a) {
#pragma unroll
                    for (int m = 0; m < NP; ++m)
                        tnxt[m] = taps[(a < FHH ? FHH - a - 1 : FS) * FS + m];
                    const int ylo = r - dir * a, yhi = r + dir * a;  // rows of slots FHH-a, FHH+a
                    const bool lo_ok = (ylo >= y0) && (ylo < yend);
                    const bool hi_ok = (a > 0) && (yhi >= y0) && (yhi < yend);
                    if (lo_ok || hi_ok) {
                        double2 T[TX];
#pragma unroll
                        for (int t = 0; t < TX; ++t) T[t] = make_double2(0.0, 0.0);
#pragma unroll
                        for (int m = 0; m < NP; ++m) {
                            const double tap = tcur[m];
#pragma unroll
                            for (int t = 0; t < TX; ++t) {
                                T[t].x = fma(tap, P[t][m].x, T[t].x);
                                T[t].y = fma(tap, P[t][m].y, T[t].y);
                            }
                        }
                        if (lo_ok) {
#pragma unroll
                            for (int t = 0; t < TX; ++t) {
                                ring[ROT ? (ph + FHH - a) % FS : FHH - a][t].x += T[t].x;
                                ring[ROT ? (ph + FHH - a) % FS : FHH - a][t].y += T[t].y;
                            }
                        }
                        if (hi_ok) {
#pragma unroll
                            for (int t = 0; t < TX; ++t) {
                                ring[ROT ? (ph + FHH + a) % FS : FHH + a][t].x += T[t].x;
                                ring[ROT ? (ph + FHH + a) % FS : FHH + a][t].y += T[t].y;
                            }
                        }
                    }
#pragma unroll
                    for (int m = 0; m < NP; ++m) tcur[m] = tnxt[m];
                }
            } else {
#pragma unroll
                for (int k = 0; k < FS; ++k) {
#pragma unroll
                    for (int m = 0; m < NP; ++m) tnxt[m] = taps[(k + 1) * FS + m];
                    const int oy = r - FHH + k;
                    if (oy >= y0 && oy < yend) {
#pragma unroll
                        for (int m = 0; m < NP; ++m) {
                            const double tap = tcur[m];
#pragma unroll
                            for (int t = 0; t < TX; ++t) {
                                double2 x;
                                if constexpr (SYMX) {
                                    x = P[t][m];
                                } else {
                                    x = row[t + FS - 1 - m];
                                }
                                ring[k][t].x = fma(tap, x.x, ring[k][t].x);
                                ring[k][t].y = fma(tap, x.y, ring[k][t].y);
                            }
                        }
                    }
#pragma unroll
                    for (int m = 0; m < NP; ++m) tcur[m] = tnxt[m];
                }
            }
        }
        if constexpr (STAMP) {
            t3 = stamp();
            if (r >= 0 && r < A.H) acc_math += t3 - t2;
        }
        const int oy0 = r - dir * FHH;  // slot 0 has received its last tap row
        if (oy0 >= y0 && oy0 < yend) {
            if constexpr (FUSE) {
                // Spectral (LSF) pass on the finished row before it is stored:
                // FSF and LSF act on different axes and commute.  The strip's
                // Dp-channel spectrum goes to a wave-private LDS buffer with a
                // circular halo of LSF_RL channels; each lane reads its aligned
                // window of 2*LSF_RL+2 channels back and applies the taps.
                constexpr int RL = LSF_RL;
                const int N = A.Dp;
                double *buf = s_spec + (size_t)s * TX * (N + 2 * RL);
#pragma unroll
                for (int t = 0; t < TX; ++t) {
                    double *bt = buf + t * (N + 2 * RL);
                    const double2 v = ring[ROT ? ph % FS : 0][t];
                    *reinterpret_cast<double2 *>(bt + RL + 2 * zl) = v;
                    if (2 * zl < RL) *reinterpret_cast<double2 *>(bt + N + RL + 2 * zl) = v;
                    if (2 * zl >= N - RL) *reinterpret_cast<double2 *>(bt + RL + 2 * zl - N) = v;
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // re-read the 17 weights here (opaque scalar zero): hoisted out of
                // the row loop they would pin 34 SGPRs for the whole kernel
                int opq_s = 0;
                asm volatile("" : "+s"(opq_s));
                const double *wl = A.lsf_dense + opq_s;
#pragma unroll
                for (int t = 0; t < TX; ++t) {
                    const double *bt = buf + t * (N + 2 * RL) + 2 * zl;
                    // streaming form: one 16-byte window read live at a time
                    //   acc.x = sum_j wl[j] w[j],  acc.y = sum_j wl[j] w[j+1]
                    double2 acc = make_double2(0.0, 0.0);
#pragma unroll
                    for (int j = 0; j < RL + 1; ++j) {
                        const double2 p = *reinterpret_cast<const double2 *>(bt + 2 * j);
                        // p = (w[2j], w[2j+1])
                        if (2 * j <= 2 * RL) acc.x = fma(wl[2 * j], p.x, acc.x);
                        if (2 * j + 1 <= 2 * RL) acc.x = fma(wl[2 * j + 1], p.y, acc.x);
                        if (2 * j - 1 >= 0) acc.y = fma(wl[2 * j - 1], p.x, acc.y);
                        if (2 * j <= 2 * RL) acc.y = fma(wl[2 * j], p.y, acc.y);
                    }
                    ring[ROT ? ph % FS : 0][t] = acc;
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_wave_barrier();
            }
#pragma unroll
            for (int t = 0; t < TX; ++t) {
                const int xo = x0 + t;
                if (xo < A.W) {
                    const long o = (long)oy0 * rowstride + (long)xo * A.Dp + 2 * zl;
                    double2 v = ring[ROT ? ph % FS : 0][t];
                    if (A.data) {
                        const double2 d = *reinterpret_cast<const double2 *>(A.data + o);
                        v.x = d.x - v.x;
                        v.y = d.y - v.y;
                    }
                    *reinterpret_cast<double2 *>(out + o) = v;
                }
            }
        }
        if constexpr (ROT) {
            // the finished slot becomes the newest one of the next phase
#pragma unroll
            for (int t = 0; t < TX; ++t) ring[ph % FS][t] = make_double2(0.0, 0.0);
        } else {
#pragma unroll
            for (int k = 0; k < FS - 1; ++k)
#pragma unroll
                for (int t = 0; t < TX; ++t) ring[k][t] = ring[k + 1][t];
#pragma unroll
            for (int t = 0; t < TX; ++t) ring[FS - 1][t] = make_double2(0.0, 0.0);
        }
        r += dir;
        if constexpr (STAMP) acc_tail += stamp() - t3;
    }
    if constexpr (STAMP) {
        const unsigned long long t_end = stamp();
        if ((threadIdx.x & 63) == 0 && A.dbg) {
            unsigned long long *d = A.dbg + ((size_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6)) * 8;
            d[0] = acc_issue;
            d[1] = acc_wait;
            d[2] = acc_math;
            d[3] = acc_tail;
            d[4] = t_end - t_begin;
            d[5] = (unsigned long long)nsteps;
            d[6] = t_begin;
            d[7] = t_end;
        }
    }
}

// Column march for an FSF that is an outer product, fsf[k][m] == u[k] * v[m] to
// rounding -- every Gaussian FSF with pa = 0 (the reference's MUSE default,
// lib/instruments.py:95-107, lib/spread_functions.py:94-131) is.  Same strips,
// ring and block order as k_spatial_march, but an input row is first reduced
// along x, X[t] = sum_m v[m] * row[t + FS - 1 - m], and X then feeds the FS ring
// slots with u[k]: 2*FS FMAs per output instead of FS*FS -- at FS = 11 the pass
// is bound by HBM, not by FP64.  The result differs from the 2-D sum by rounding
// only (the host checks |fsf - u v^T| <= 8 eps max|fsf| before choosing this).
template <int NT, int FS, int TX, bool UNI>
__global__ __launch_bounds__(NT, 2) void k_spatial_sep(SpatialArgs A, const double *__restrict__ in,
                                                       double *__restrict__ out, int HY) {
    constexpr int FHH = (FS - 1) / 2;
    constexpr int NR = TX + FS - 1;  // inputs per row
    __shared__ double s_uv[2 * FS];  // u (tap rows, y) | v (tap columns, x)
    for (int i = threadIdx.x; i < 2 * FS; i += NT) s_uv[i] = A.sep_uv[i];
    __syncthreads();

    const int S = NT / A.HL;
    int s = threadIdx.x / A.HL;
    const int zl = threadIdx.x - s * A.HL;
    if constexpr (UNI) s = __builtin_amdgcn_readfirstlane(s);
    const int nxs = (A.W + TX - 1) / TX;
    const int nys = (A.H + HY - 1) / HY;
    int blk = blockIdx.x;  // XCD-aware block order, as in k_spatial_march
    if (A.xcd_remap) {
        const int nb = gridDim.x, q = nb / 8, rm = nb % 8, xcd = blk % 8;
        blk = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + blk / 8;
    }
    const long item = (long)blk * S + s;
    if (s >= S || item >= (long)nxs * nys) return;
    const int ys = (int)(item / nxs);
    const int x0 = (int)(item - (long)ys * nxs) * TX;
    const int y0 = ys * HY;
    const int yend = min(y0 + HY, A.H);
    const long rowstride = (long)A.W * A.Dp;
    const bool xin = (x0 - FHH >= 0) && (x0 + TX - 1 + FHH < A.W);

    double2 ring[FS][TX];
#pragma unroll
    for (int k = 0; k < FS; ++k)
#pragma unroll
        for (int t = 0; t < TX; ++t) ring[k][t] = make_double2(0.0, 0.0);

    // the march is unrolled by FS and ring slot k of phase ph lives in physical
    // register (ph + k) % FS: the slot of an output row never moves
    const int nsteps = (yend - y0) + 2 * FHH;
    int r = y0 - FHH;
    for (int sbase = 0; sbase < nsteps; sbase += FS)
#pragma unroll
    for (int ph = 0; ph < FS; ++ph) {
        if (sbase + ph >= nsteps) continue;
        if (r >= 0 && r < A.H) {
            const double *base = in + (long)r * rowstride + (long)(x0 - FHH) * A.Dp + 2 * zl;
            double2 row[NR];
            if (xin) {
#pragma unroll
                for (int i = 0; i < NR; ++i)
                    row[i] = *reinterpret_cast<const double2 *>(base + (long)i * A.Dp);
            } else {
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    const int xx = x0 - FHH + i;
                    row[i] = (xx >= 0 && xx < A.W)
                                 ? *reinterpret_cast<const double2 *>(base + (long)i * A.Dp)
                                 : make_double2(0.0, 0.0);
                }
            }
            // taps re-read from LDS every step (opaque zero: no 2*FS pinned registers)
            int opq = 0;
            asm volatile("" : "+v"(opq));
            const double *uv = s_uv + opq;
            double2 X[TX];
#pragma unroll
            for (int t = 0; t < TX; ++t) {
                const double v0 = uv[FS];
                X[t].x = v0 * row[t + FS - 1].x;
                X[t].y = v0 * row[t + FS - 1].y;
            }
#pragma unroll
            for (int m = 1; m < FS; ++m) {
                const double vm = uv[FS + m];
#pragma unroll
                for (int t = 0; t < TX; ++t) {
                    X[t].x = fma(vm, row[t + FS - 1 - m].x, X[t].x);
                    X[t].y = fma(vm, row[t + FS - 1 - m].y, X[t].y);
                }
            }
#pragma unroll
            for (int k = 0; k < FS; ++k) {
                const int oy = r - FHH + k;
                if (oy >= y0 && oy < yend) {
                    const double uk = uv[k];
#pragma unroll
                    for (int t = 0; t < TX; ++t) {
                        ring[(ph + k) % FS][t].x = fma(uk, X[t].x, ring[(ph + k) % FS][t].x);
                        ring[(ph + k) % FS][t].y = fma(uk, X[t].y, ring[(ph + k) % FS][t].y);
                    }
                }
            }
        }
        const int oy0 = r - FHH;  // slot 0 has received its last tap row
        if (oy0 >= y0 && oy0 < yend) {
#pragma unroll
            for (int t = 0; t < TX; ++t) {
                const int xo = x0 + t;
                if (xo < A.W) {
                    const long o = (long)oy0 * rowstride + (long)xo * A.Dp + 2 * zl;
                    double2 v = ring[ph % FS][t];
                    if (A.data) {
                        const double2 d = *reinterpret_cast<const double2 *>(A.data + o);
                        v.x = d.x - v.x;
                        v.y = d.y - v.y;
                    }
                    *reinterpret_cast<double2 *>(out + o) = v;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < TX; ++t) ring[ph % FS][t] = make_double2(0.0, 0.0);
        r += 1;
    }
}

// LSF x FSF in ONE pass for an outer-product FSF: k_spatial_sep whose finished
// output row goes through the dense LSF before it is stored (FSF and LSF act on
// different axes and commute).  The strip's Dp-channel spectrum passes through a
// wave-private LDS buffer with a circular halo of LSF_RL channels and each lane
// applies the taps to its aligned window.  Needs a power-of-two depth whose
// spectrum fits a wavefront and taps within +-LSF_RL (c->lsf_fusable).
// The row loop is NOT unrolled and the ring shifts by register moves: the
// rotating-ring form replicates the epilogue FS times and the compiler then
// overlaps the next row's 13 loads with it -- 256 VGPRs and 273 spills (115 us).
template <int NT, int FS, int TX, bool UNI>
__global__ __launch_bounds__(NT, 2) void k_spatial_sep_lsf(SpatialArgs A, const double *__restrict__ in,
                                                           double *__restrict__ out, int HY) {
    constexpr int FHH = (FS - 1) / 2;
    constexpr int NR = TX + FS - 1;
    constexpr int RL = LSF_RL;
    __shared__ double s_uv[2 * FS];
    extern __shared__ double s_spec[];  // [strip][TX][Dp + 2*RL]
    for (int i = threadIdx.x; i < 2 * FS; i += NT) s_uv[i] = A.sep_uv[i];
    __syncthreads();

    const int S = NT / A.HL;
    int s = threadIdx.x / A.HL;
    const int zl = threadIdx.x - s * A.HL;
    if constexpr (UNI) s = __builtin_amdgcn_readfirstlane(s);
    const int nxs = (A.W + TX - 1) / TX;
    const int nys = (A.H + HY - 1) / HY;
    int blk = blockIdx.x;
    if (A.xcd_remap) {
        const int nb = gridDim.x, q = nb / 8, rm = nb % 8, xcd = blk % 8;
        blk = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + blk / 8;
    }
    const long item = (long)blk * S + s;
    if (s >= S || item >= (long)nxs * nys) return;
    const int ys = (int)(item / nxs);
    const int x0 = (int)(item - (long)ys * nxs) * TX;
    const int y0 = ys * HY;
    const int yend = min(y0 + HY, A.H);
    const long rowstride = (long)A.W * A.Dp;
    const bool xin = (x0 - FHH >= 0) && (x0 + TX - 1 + FHH < A.W);
    const int N = A.Dp;
    double *buf = s_spec + (size_t)s * TX * (N + 2 * RL);

    double2 ring[FS][TX];
#pragma unroll
    for (int k = 0; k < FS; ++k)
#pragma unroll
        for (int t = 0; t < TX; ++t) ring[k][t] = make_double2(0.0, 0.0);

    // Software pipeline without extra registers: a row is reduced along x as soon
    // as it has arrived (X), which frees its registers, and the NEXT row's loads
    // are issued right then -- they fly while X feeds the ring, the finished row
    // goes through the LSF and the ring shifts.
    const int nsteps = (yend - y0) + 2 * FHH;
    double2 row[NR];
    auto load_row = [&](int r) {
        if (r >= 0 && r < A.H) {
            const double *base = in + (long)r * rowstride + (long)(x0 - FHH) * A.Dp + 2 * zl;
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int xx = x0 - FHH + i;
                row[i] = (xin || (xx >= 0 && xx < A.W))
                             ? *reinterpret_cast<const double2 *>(base + (long)i * A.Dp)
                             : make_double2(0.0, 0.0);
            }
        }
    };
    load_row(y0 - FHH);
#pragma unroll 1
    for (int step = 0; step < nsteps; ++step) {
        const int r = y0 - FHH + step;
        const bool live = r >= 0 && r < A.H;
        double2 X[TX];
        if (live) {
#pragma unroll
            for (int t = 0; t < TX; ++t) {
                const double v0 = s_uv[FS];
                X[t].x = v0 * row[t + FS - 1].x;
                X[t].y = v0 * row[t + FS - 1].y;
            }
#pragma unroll
            for (int m = 1; m < FS; ++m) {
                const double vm = s_uv[FS + m];
#pragma unroll
                for (int t = 0; t < TX; ++t) {
                    X[t].x = fma(vm, row[t + FS - 1 - m].x, X[t].x);
                    X[t].y = fma(vm, row[t + FS - 1 - m].y, X[t].y);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the next row's loads behind the x reduction
        if (step + 1 < nsteps) load_row(r + 1);
        __builtin_amdgcn_sched_barrier(0);  // ... and ahead of everything below
        if (live) {
#pragma unroll
            for (int k = 0; k < FS; ++k) {
                const int oy = r - FHH + k;
                if (oy >= y0 && oy < yend) {
                    const double uk = s_uv[k];
#pragma unroll
                    for (int t = 0; t < TX; ++t) {
                        ring[k][t].x = fma(uk, X[t].x, ring[k][t].x);
                        ring[k][t].y = fma(uk, X[t].y, ring[k][t].y);
                    }
                }
            }
        }
        const int oy0 = r - FHH;  // slot 0 has received its last tap row
        if (oy0 >= y0 && oy0 < yend) {
#pragma unroll
            for (int t = 0; t < TX; ++t) {
                double *bt = buf + t * (N + 2 * RL);
                const double2 v = ring[0][t];
                *reinterpret_cast<double2 *>(bt + RL + 2 * zl) = v;
                if (2 * zl < RL) *reinterpret_cast<double2 *>(bt + N + RL + 2 * zl) = v;
                if (2 * zl >= N - RL) *reinterpret_cast<double2 *>(bt + RL + 2 * zl - N) = v;
            }
            __builtin_amdgcn_wave_barrier();  // wave-private buffer: LDS is in order per wave
#pragma unroll
            for (int t = 0; t < TX; ++t) {
                const int xo = x0 + t;
                const double *bt = buf + t * (N + 2 * RL) + 2 * zl;
                //   acc.x = sum_j wl[j] w[j],  acc.y = sum_j wl[j] w[j+1]
                double2 acc = make_double2(0.0, 0.0);
#pragma unroll
                for (int j = 0; j < RL + 1; ++j) {
                    const double2 p = *reinterpret_cast<const double2 *>(bt + 2 * j);
                    if (2 * j <= 2 * RL) acc.x = fma(A.lsf_dense[2 * j], p.x, acc.x);
                    if (2 * j + 1 <= 2 * RL) acc.x = fma(A.lsf_dense[2 * j + 1], p.y, acc.x);
                    if (2 * j - 1 >= 0) acc.y = fma(A.lsf_dense[2 * j - 1], p.x, acc.y);
                    if (2 * j <= 2 * RL) acc.y = fma(A.lsf_dense[2 * j], p.y, acc.y);
                }
                if (xo < A.W) {
                    const long o = (long)oy0 * rowstride + (long)xo * A.Dp + 2 * zl;
                    if (A.data) {
                        const double2 d = *reinterpret_cast<const double2 *>(A.data + o);
                        acc.x = d.x - acc.x;
                        acc.y = d.y - acc.y;
                    }
                    *reinterpret_cast<double2 *>(out + o) = acc;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int k = 0; k < FS - 1; ++k)
#pragma unroll
            for (int t = 0; t < TX; ++t) ring[k][t] = ring[k + 1][t];
#pragma unroll
        for (int t = 0; t < TX; ++t) ring[FS - 1][t] = make_double2(0.0, 0.0);
    }
}

// Software-pipelined column march for FSFs that are mirror-symmetric in x AND
// y, one strip per wavefront (HL multiple of 64).  Same ring / symmetry algebra
// as k_spatial_march<SYMX, SYMY>; in addition the NEXT input row is loaded while
// this row is being consumed: its TX+FS-1 loads are issued in small groups
// between the FHH+1 tap-row blocks, so the memory pipe (64 B/clk per CU, shared
// by 8 wavefronts) digests them under the FMAs instead of in a burst that stalls
// every wave of the CU at the same time (in-kernel stamps of the burst form:
// issue 1300 + wait 680 cycles of a 6500-cycle step).  TX = 2 columns per
// thread keeps ring + folded row + next row within 256 VGPRs (2 waves/SIMD).
template <int NT, int FS, int TX>
__global__ __launch_bounds__(NT, 2) void k_spatial_march_pf(SpatialArgs A,
                                                            const double *__restrict__ in,
                                                            double *__restrict__ out, int HY) {
    constexpr int FHH = (FS - 1) / 2;
    constexpr int NP = FHH + 1;
    constexpr int NR = TX + FS - 1;
    constexpr int CH = (NR + NP - 1) / NP;  // loads issued per tap-row block
    __shared__ double s_taps[NP * NP + NP];
    for (int i = threadIdx.x; i < NP * NP + NP; i += NT) {
        const int k = i / NP, m = i - k * NP;
        s_taps[i] = k < NP ? A.fsf[k * FS + m] : 0.0;  // rows 0..FHH, columns 0..FHH
    }
    __syncthreads();

    const int S = NT / A.HL;
    const int s = __builtin_amdgcn_readfirstlane(threadIdx.x / A.HL);
    const int zl = threadIdx.x - s * A.HL;
    const int nxs = (A.W + TX - 1) / TX;
    const int nys = (A.H + HY - 1) / HY;
    int blk = blockIdx.x;
    if (A.xcd_remap) {
        const int nb = gridDim.x, q = nb / 8, rm = nb % 8, xcd = blk % 8;
        blk = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + blk / 8;
    }
    const long item = (long)blk * S + s;
    if (s >= S || item >= (long)nxs * nys) return;
    const int ys = (int)(item / nxs);
    const int x0 = (int)(item - (long)ys * nxs) * TX;
    const int y0 = ys * HY;
    const int yend = min(y0 + HY, A.H);
    const long rowstride = (long)A.W * A.Dp;
    const int dir = (A.alt_dir && (ys & 1)) ? -1 : 1;
    const int nsteps = (yend - y0) + 2 * FHH;
    const double *colbase = in + (long)(x0 - FHH) * A.Dp + 2 * zl;

    // column i of input row rr, or zero outside the cube (all tests wave-uniform)
    auto load_col = [&](int rr, int i) -> double2 {
        const int xx = x0 - FHH + i;
        if (rr >= 0 && rr < A.H && xx >= 0 && xx < A.W)
            return *reinterpret_cast<const double2 *>(colbase + (long)rr * rowstride +
                                                      (long)i * A.Dp);
        return make_double2(0.0, 0.0);
    };

    double2 ring[FS][TX];
#pragma unroll
    for (int k = 0; k < FS; ++k)
#pragma unroll
        for (int t = 0; t < TX; ++t) ring[k][t] = make_double2(0.0, 0.0);

    int r = dir > 0 ? y0 - FHH : yend - 1 + FHH;
    double2 cur[NR], nxt[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) cur[i] = load_col(r, i);

    for (int step = 0; step < nsteps; ++step, r += dir) {
        // fold the row: P[t][m] = cur[t+m] + cur[t+FS-1-m] (m < FHH), centre cur[t+FHH]
        double2 P[TX][NP];
#pragma unroll
        for (int t = 0; t < TX; ++t) {
#pragma unroll
            for (int m = 0; m < FHH; ++m) {
                P[t][m].x = cur[t + m].x + cur[t + FS - 1 - m].x;
                P[t][m].y = cur[t + m].y + cur[t + FS - 1 - m].y;
            }
            P[t][FHH] = cur[t + FHH];
        }
        const bool more = step + 1 < nsteps;
        const int rn = r + dir;
        int opq = 0;
        asm volatile("" : "+v"(opq));  // keep the tap reads inside the loop
        const double *taps = s_taps + opq;
#pragma unroll
        for (int a = 0; a <= FHH; ++a) {
            // a few loads of the next row, then the tap-row block they hide under
#pragma unroll
            for (int i = a * CH; i < (a + 1) * CH && i < NR; ++i)
                nxt[i] = more ? load_col(rn, i) : make_double2(0.0, 0.0);
            const int ylo = r - dir * a, yhi = r + dir * a;  // rows of slots FHH-a, FHH+a
            const bool lo_ok = (ylo >= y0) && (ylo < yend);
            const bool hi_ok = (a > 0) && (yhi >= y0) && (yhi < yend);
            if (lo_ok || hi_ok) {
                double2 T[TX];
#pragma unroll
                for (int t = 0; t < TX; ++t) T[t] = make_double2(0.0, 0.0);
#pragma unroll
                for (int m = 0; m < NP; ++m) {
                    const double tap = taps[(FHH - a) * NP + m];
#pragma unroll
                    for (int t = 0; t < TX; ++t) {
                        T[t].x = fma(tap, P[t][m].x, T[t].x);
                        T[t].y = fma(tap, P[t][m].y, T[t].y);
                    }
                }
                if (lo_ok) {
#pragma unroll
                    for (int t = 0; t < TX; ++t) {
                        ring[FHH - a][t].x += T[t].x;
                        ring[FHH - a][t].y += T[t].y;
                    }
                }
                if (hi_ok) {
#pragma unroll
                    for (int t = 0; t < TX; ++t) {
                        ring[FHH + a][t].x += T[t].x;
                        ring[FHH + a][t].y += T[t].y;
                    }
                }
            }
        }
        const int oy0 = r - dir * FHH;  // slot 0 has received its last tap row
        if (oy0 >= y0 && oy0 < yend) {
#pragma unroll
            for (int t = 0; t < TX; ++t) {
                const int xo = x0 + t;
                if (xo < A.W) {
                    const long o = (long)oy0 * rowstride + (long)xo * A.Dp + 2 * zl;
                    double2 v = ring[0][t];
                    if (A.data) {
                        const double2 d = *reinterpret_cast<const double2 *>(A.data + o);
                        v.x = d.x - v.x;
                        v.y = d.y - v.y;
                    }
                    *reinterpret_cast<double2 *>(out + o) = v;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < FS - 1; ++k)
#pragma unroll
            for (int t = 0; t < TX; ++t) ring[k][t] = ring[k + 1][t];
#pragma unroll
        for (int t = 0; t < TX; ++t) ring[FS - 1][t] = make_double2(0.0, 0.0);
#pragma unroll
        for (int i = 0; i < NR; ++i) cur[i] = nxt[i];
    }
}

// One-channel-per-lane column march for x- and y-symmetric FSFs.  Same ring /
// symmetry algebra as k_spatial_march<SYMX, SYMY>, but a thread owns ONE channel
// (8-byte accesses, a spectrum spans Dp/64 wavefronts): half the registers per
// thread, hence WPS = 3-4 wavefronts per SIMD instead of 2 to hide the bursts of
// row loads behind other waves' FMAs.  Requires Dp to be a multiple of 64.
template <int NT, int FS, int TX, int WPS>
__global__ __launch_bounds__(NT, WPS) void k_spatial_march1(SpatialArgs A,
                                                            const double *__restrict__ in,
                                                            double *__restrict__ out, int HY) {
    constexpr int FHH = (FS - 1) / 2;
    constexpr int NP = FHH + 1;
    constexpr int NR = TX + FS - 1;
    __shared__ double s_taps[NP * NP + NP];
    for (int i = threadIdx.x; i < NP * NP + NP; i += NT) {
        const int k = i / NP, m = i - k * NP;
        s_taps[i] = k < NP ? A.fsf[k * FS + m] : 0.0;
    }
    __syncthreads();

    const int Dp = A.Dp;
    const int S = NT / Dp;  // strips per workgroup
    const int s = __builtin_amdgcn_readfirstlane(threadIdx.x / Dp);
    const int ch = threadIdx.x - s * Dp;
    const int nxs = (A.W + TX - 1) / TX;
    const int nys = (A.H + HY - 1) / HY;
    int blk = blockIdx.x;
    if (A.xcd_remap) {
        const int nb = gridDim.x, q = nb / 8, rm = nb % 8, xcd = blk % 8;
        blk = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + blk / 8;
    }
    const long item = (long)blk * S + s;
    if (s >= S || item >= (long)nxs * nys) return;
    const int ys = (int)(item / nxs);
    const int x0 = (int)(item - (long)ys * nxs) * TX;
    const int y0 = ys * HY;
    const int yend = min(y0 + HY, A.H);
    const long rowstride = (long)A.W * Dp;
    const int dir = (A.alt_dir && (ys & 1)) ? -1 : 1;
    const int nsteps = (yend - y0) + 2 * FHH;
    const double *colbase = in + (long)(x0 - FHH) * Dp + ch;

    double ring[FS][TX];
#pragma unroll
    for (int k = 0; k < FS; ++k)
#pragma unroll
        for (int t = 0; t < TX; ++t) ring[k][t] = 0.0;

    int r = dir > 0 ? y0 - FHH : yend - 1 + FHH;
    // the march is unrolled by FS: ring slot k of phase ph lives in physical
    // register (ph + k) % FS, so the ring never moves
    for (int sbase = 0; sbase < nsteps; sbase += FS)
#pragma unroll
    for (int ph = 0; ph < FS; ++ph) {
        const int step = sbase + ph;
        if (step >= nsteps) continue;
        if (r >= 0 && r < A.H) {
            double row[NR];
            const double *base = colbase + (long)r * rowstride;
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int xx = x0 - FHH + i;
                row[i] = (xx >= 0 && xx < A.W) ? base[(long)i * Dp] : 0.0;
            }
            double P[TX][NP];
#pragma unroll
            for (int t = 0; t < TX; ++t) {
#pragma unroll
                for (int m = 0; m < FHH; ++m) P[t][m] = row[t + m] + row[t + FS - 1 - m];
                P[t][FHH] = row[t + FHH];
            }
            int opq = 0;
            asm volatile("" : "+v"(opq));  // keep the tap reads inside the loop
            const double *taps = s_taps + opq;
#pragma unroll
            for (int a = 0; a <= FHH; ++a) {
                const int ylo = r - dir * a, yhi = r + dir * a;  // rows of slots FHH-a, FHH+a
                const bool lo_ok = (ylo >= y0) && (ylo < yend);
                const bool hi_ok = (a > 0) && (yhi >= y0) && (yhi < yend);
                if (lo_ok || hi_ok) {
                    double T[TX];
#pragma unroll
                    for (int t = 0; t < TX; ++t) T[t] = 0.0;
#pragma unroll
                    for (int m = 0; m < NP; ++m) {
                        const double tap = taps[(FHH - a) * NP + m];
#pragma unroll
                        for (int t = 0; t < TX; ++t) T[t] = fma(tap, P[t][m], T[t]);
                    }
                    if (lo_ok) {
#pragma unroll
                        for (int t = 0; t < TX; ++t) ring[(ph + FHH - a) % FS][t] += T[t];
                    }
                    if (hi_ok) {
#pragma unroll
                        for (int t = 0; t < TX; ++t) ring[(ph + FHH + a) % FS][t] += T[t];
                    }
                }
            }
        }
        const int oy0 = r - dir * FHH;  // slot 0 has received its last tap row
        if (oy0 >= y0 && oy0 < yend) {
#pragma unroll
            for (int t = 0; t < TX; ++t) {
                const int xo = x0 + t;
                if (xo < A.W) {
                    const long o = (long)oy0 * rowstride + (long)xo * Dp + ch;
                    double v = ring[ph % FS][t];
                    if (A.data) v = A.data[o] - v;
                    out[o] = v;
                }
            }
        }
#pragma unroll
        for (int t = 0; t < TX; ++t) ring[ph % FS][t] = 0.0;  // newest slot of the next phase
        r += dir;
    }
}

// Any-size fallback: one output z-pair per thread, loops over all taps.
template <int NT>
__global__ __launch_bounds__(NT) void k_spatial_generic(SpatialArgs A,
                                                        const double *__restrict__ in,
                                                        double *__restrict__ out) {
    const int S = NT / A.HL;
    const int s = threadIdx.x / A.HL, zl = threadIdx.x - s * A.HL;
    const long sp = (long)blockIdx.x * S + s;
    if (s >= S || sp >= (long)A.H * A.W) return;
    const int y = (int)(sp / A.W), x = (int)(sp - (long)y * A.W);
    const int fhh = (A.fh - 1) / 2, fhw = (A.fw - 1) / 2;
    double2 acc = make_double2(0.0, 0.0);
    for (int j = 0; j < A.fh; ++j) {
        const int yy = y - j + fhh;
        if (yy < 0 || yy >= A.H) continue;
        for (int i = 0; i < A.fw; ++i) {
            const int xx = x - i + fhw;
            if (xx < 0 || xx >= A.W) continue;
            const double tap = A.fsf[j * A.fw + i];
            const double2 v =
                *reinterpret_cast<const double2 *>(in + ((long)yy * A.W + xx) * A.Dp + 2 * zl);
            acc.x = fma(tap, v.x, acc.x);
            acc.y = fma(tap, v.y, acc.y);
        }
    }
    const long o = sp * A.Dp + 2 * zl;
    if (A.data) {
        const double2 d = *reinterpret_cast<const double2 *>(A.data + o);
        acc.x = d.x - acc.x;
        acc.y = d.y - acc.y;
    }
    *reinterpret_cast<double2 *>(out + o) = acc;
}

// ------------------------------------------------------------------------- //
// z-major convolution kernels: the cube in the REFERENCE layout (D, H, W),     //
// x fastest (lib/run.py:146-149), lanes along x.  d3d_convolve works there      //
// without any layout change.                                                   //
// ------------------------------------------------------------------------- //

// Spectral pass, z-major: a thread owns one spaxel (consecutive lanes =
// consecutive x: every access 8 B/lane contiguous) and marches along z with the
// 2*LSF_RL+1 channels around the current one in registers:
//   out[k] = sum_j wl[j] * in[(k + j - RL) mod D]       (D a power of two)
// The window rotates at compile time (the march is unrolled by its length).
static __global__ __launch_bounds__(256) void k_spectral_z(int D, long HW, const double *__restrict__ wl,
                                                    const double *__restrict__ in,
                                                    double *__restrict__ out) {
    constexpr int RL = LSF_RL, NW = 2 * LSF_RL + 1;
    const long sp = (long)blockIdx.x * 256 + threadIdx.x;
    if (sp >= HW) return;
    double w[NW], t[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        t[j] = wl[j];
        w[j] = in[(long)((j - RL) & (D - 1)) * HW + sp];  // channels -RL .. +RL around 0
    }
    // invariant at output channel k (k = base + ph): w[(ph + j) % NW] = in[k + j - RL]
    for (int base = 0; base < D; base += NW) {
#pragma unroll
        for (int ph = 0; ph < NW; ++ph) {
            const int k = base + ph;
            if (k < D) {
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < NW; ++j) acc = fma(t[j], w[(ph + j) % NW], acc);
                out[(long)k * HW + sp] = acc;
                // slot of in[k - RL] is free: load in[k + RL + 1] for the next channel
                w[ph % NW] = in[(long)((k + RL + 1) & (D - 1)) * HW + sp];
            }
        }
    }
}

// Spatial pass, z-major, for FSFs that are mirror-symmetric in x and y.  A
// wavefront owns 64 consecutive columns of one channel image and marches down HY
// rows; each input row segment (64 + FS-1 values) goes through a wave-private
// LDS row, every lane folds its FS neighbours (x symmetry), forms the FHH+1
// tap-row dot products once and adds them to the two ring slots FHH-a / FHH+a
// (y symmetry).  The ring (FS doubles per thread) rotates at compile time.  An
// input value is loaded once per tile (plus the halo), the kernel needs ~60
// VGPRs: 8 wavefronts per SIMD, HBM-bound.
// (Asked for six wavefronts per SIMD the compiler needs 68 instead of 122 VGPRs at FS = 11 without a
// spill -- 9 and 15 taps spill there and keep four: 300x300x128 109 -> 100 us per convolution,
// 64^3 25 -> 21.)
template <int FS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((FS == 9 || FS == 15) ? 4 : 6, 8))) void k_spatial_z(int D, int H, int W, int HY,
                                                   const double *__restrict__ fsf,
                                                   const double *__restrict__ in,
                                                   double *__restrict__ out) {
    constexpr int FHH = (FS - 1) / 2, NP = FHH + 1;
    constexpr int ROWLEN = 64 + FS - 1;
    __shared__ double s_row[4][ROWLEN + 1];
    // wave index made scalar: every row/strip test below is then a scalar branch
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int nxt = (W + 63) / 64, nys = (H + HY - 1) / HY;
    const long item = (long)blockIdx.x * 4 + wave;
    if (item >= (long)D * nys * nxt) return;
    const int z = (int)(item / ((long)nys * nxt));
    const int rem = (int)(item - (long)z * nys * nxt);
    const int ys = rem / nxt, xt = rem - ys * nxt;
    const int x = xt * 64 + lane;
    const int y0 = ys * HY, yend = min(y0 + HY, H);
    const double *img = in + (long)z * H * W;
    double *oimg = out + (long)z * H * W;
    double *rowbuf = s_row[wave];

    double tap[NP][NP];  // rows 0..FHH, columns 0..FHH of the FSF (the rest by symmetry)
#pragma unroll
    for (int k = 0; k < NP; ++k)
#pragma unroll
        for (int m = 0; m < NP; ++m) tap[k][m] = fsf[k * FS + m];

    double ring[FS];
#pragma unroll
    for (int k = 0; k < FS; ++k) ring[k] = 0.0;

    const int nsteps = (yend - y0) + 2 * FHH;
    // one row segment [xt*64 - FHH, xt*64 + 63 + FHH]: lane l holds element l,
    // lanes < FS-1 also element l + 64 (zero outside the image / the cube)
    const int xa = xt * 64 - FHH + lane, xb = xa + 64;
    const bool a_in = xa >= 0 && xa < W, b_in = lane < FS - 1 && xb >= 0 && xb < W;
    auto fetch = [&](int r, double &va, double &vb) {
        va = 0.0;
        vb = 0.0;
        if (r >= 0 && r < H) {
            if (a_in) va = img[(long)r * W + xa];
            if (b_in) vb = img[(long)r * W + xb];
        }
    };
    double va, vb;
    fetch(y0 - FHH, va, vb);
    // step s reads input row r = y0 - FHH + s.  The slot of output row
    // (r - FHH + k) is (ph + k) % FS at phase ph = s % FS: constant per output
    // row, so the ring never moves (the march is unrolled by FS).
    for (int base = 0; base < nsteps; base += FS) {
#pragma unroll
        for (int ph = 0; ph < FS; ++ph) {
            const int step = base + ph;
            if (step < nsteps) {
                const int r = y0 - FHH + step;
                // this row through the wave-private LDS row; next row's loads fly
                // (four rows in flight instead of one changed nothing: 126 against 119 us per
                // convolution -- the march waits on its own dependent chain, not on memory)
                rowbuf[lane] = va;
                if (lane < FS - 1) rowbuf[lane + 64] = vb;
                __builtin_amdgcn_wave_barrier();
                double P[NP];
#pragma unroll
                for (int m = 0; m < FHH; ++m) P[m] = rowbuf[lane + m] + rowbuf[lane + FS - 1 - m];
                P[FHH] = rowbuf[lane + FHH];
                __builtin_amdgcn_wave_barrier();
                fetch(r + 1, va, vb);
                if (r >= 0 && r < H) {
                    if ((r - FHH >= y0) && (r + FHH < yend)) {
                        // steady state: every slot is live, no range tests; the
                        // FHH+1 dot products advance together (independent chains)
                        double T[NP];
#pragma unroll
                        for (int a = 0; a <= FHH; ++a) T[a] = tap[FHH - a][0] * P[0];
#pragma unroll
                        for (int m = 1; m < NP; ++m)
#pragma unroll
                            for (int a = 0; a <= FHH; ++a) T[a] = fma(tap[FHH - a][m], P[m], T[a]);
#pragma unroll
                        for (int a = 0; a <= FHH; ++a) {
                            ring[(ph + FHH - a) % FS] += T[a];
                            if (a > 0) ring[(ph + FHH + a) % FS] += T[a];
                        }
                    } else {
#pragma unroll
                        for (int a = 0; a <= FHH; ++a) {
                            const int ylo = r - a, yhi = r + a;
                            const bool lo_ok = ylo >= y0 && ylo < yend;
                            const bool hi_ok = a > 0 && yhi >= y0 && yhi < yend;
                            if (lo_ok || hi_ok) {
                                double T = 0.0;
#pragma unroll
                                for (int m = 0; m < NP; ++m) T = fma(tap[FHH - a][m], P[m], T);
                                // output row r-a sits in slot k = FHH - a, row r+a in k = FHH + a
                                if (lo_ok) ring[(ph + FHH - a) % FS] += T;
                                if (hi_ok) ring[(ph + FHH + a) % FS] += T;
                            }
                        }
                    }
                }
                const int oy0 = r - FHH;  // slot k = 0 is complete
                if (oy0 >= y0 && oy0 < yend && x < W) oimg[(long)oy0 * W + x] = ring[ph % FS];
                ring[ph % FS] = 0.0;  // becomes slot k = FS-1 of the next step
            }
        }
    }
}

// ------------------------------------------------------------------------- //
// chi2 map                                                                    //
// ------------------------------------------------------------------------- //

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Sum over the wavefront with DPP row shifts / broadcasts instead of LDS permutes
// (no LDS round trip per step); the total lands in lane 63 and is broadcast.
// Another association than wave_sum: use one or the other consistently.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_add(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, ROW_MASK, BANK_MASK, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, BANK_MASK, false);
    // lanes that receive nothing (masked out or shifted in) add +0.0
    return v + __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}

// (the total is valid in lane 63 only)
__device__ __forceinline__ double wave_sum_dpp63(double v) {
    v = dpp_add<0x111, 0xf, 0xf>(v);  // row_shr:1
    v = dpp_add<0x112, 0xf, 0xf>(v);  // row_shr:2
    v = dpp_add<0x114, 0xf, 0xe>(v);  // row_shr:4
    v = dpp_add<0x118, 0xf, 0xc>(v);  // row_shr:8   -> lane 15 of each row holds the row sum
    v = dpp_add<0x142, 0xa, 0xf>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_add<0x143, 0xc, 0xf>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return v;
}

// out[sp] = 0.5 * sum_z err^2 * ivar (lib/run.py:423 per spectrum); one wave
// per spaxel.
static __global__ __launch_bounds__(256) void k_chi2_map(const double *__restrict__ err,
                                                   const double *__restrict__ ivar,
                                                   double *__restrict__ out, int HL, int Dp,
                                                   long nspax) {
    const int lane = threadIdx.x & 63;
    const long sp = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (sp >= nspax) return;
    double acc = 0.0;
    for (int zl = lane; zl < HL; zl += 64) {
        const double2 e = *reinterpret_cast<const double2 *>(err + sp * Dp + 2 * zl);
        const double2 v = *reinterpret_cast<const double2 *>(ivar + sp * Dp + 2 * zl);
        acc = fma(e.x * e.x, v.x, acc);
        acc = fma(e.y * e.y, v.y, acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) out[sp] = 0.5 * acc;
}

// Deterministic single-block sum of n doubles.
static __global__ __launch_bounds__(1024) void k_sum(const double *__restrict__ v, long n,
                                               double *__restrict__ out) {
    __shared__ double part[16];
    double acc = 0.0;
    for (long i = threadIdx.x; i < n; i += 1024) acc += v[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < 16; ++i) t += part[i];
        *out = t;
    }
}

// ------------------------------------------------------------------------- //
// MH-within-Gibbs update of one colour class (lib/run.py:367-519)            //
// ------------------------------------------------------------------------- //
//
// One workgroup per spaxel.  With e = err, v = 1/var, f = fsf over the window
// and E_old/E_new the unit-amplitude LSF-convolved lines of the current /
// proposed (c,w), everything the reference computes from full-cube
// temporaries follows from three per-channel window sums
//     A[z] = sum_pos f v e,   B[z] = sum_pos f^2 v,   C[z] = sum_pos v e^2 :
//   ar_old          = 1/2 sum_z C                                  (run.py:423)
//   ar_old - ar_new = -sum_z d A - 1/2 sum_z d^2 B, d = a (E_old - E_new)  (:426)
//   sum ek^2/var    = sum_z E^2 B,  sum ek ul/var = sum_z E (A + a E_old B) (:492-493)
// so ONE pass over the err and 1/var windows suffices, and the residual changes
// by  err += f * G[z],  G = a E_old - r E_end                      (:508-515).
//
// Write-back schemes, bit-identical in their results:
//   k_mh        immediate: pass 2 re-reads (or keeps in registers) the err
//               window and writes e + f G.  4 cube passes per colour.
//   k_mh_defer  deferred: the update is NOT written; G goes to a small side
//               buffer.  The NEXT colour's launch applies it while it streams
//               its own windows (same-colour windows tile the cube, so every
//               voxel has exactly one pending spaxel): each voxel is read once
//               and written once per colour -- the algorithmic 3 passes
//               (read err, read 1/var, write err) and no second pass.
//   k_mh_ws     deferred, wave-specialised, and with up to MH_LAYERS colours
//               pending at once: the residual is written back only every M-th
//               colour (the default path; see "Several pending layers").
//   k_mh_flow   k_mh_ws's window code under per-window dependencies, one launch
//               per sweep (opt-in).

struct MHProposal;
struct MHChainArgs {  // one chain of a batch: see MHArgs::batch
    double *err;
    const double *ivar;
    double ivar_uniform;
    double *params, *prev, *dlog;
    unsigned long long *accepted;
    double *gbuf[4];
    double min_b[3], max_b[3], amp[3];
    double ra;
    uint64_t seed;
    // (k_mh_small: the chain's tables of the sweep)
    const MHProposal *props;
    const double *ltab;
};
struct MHArgs {
    int D, Dp, HL, H, W, fh, fw, N, ntaps, npos;
    double *err;
    const double *ivar;
    double ivar_uniform;  // the constant 1/variance when the cube is uniform (k_mh_ws<.., true>)
    double *params;
    double *prev;  // [H*W*3] parameters before the last update of each spaxel, or NULL
    const double *fsf;
    const int *shift;
    const double *weight;
    double *dlog;
    unsigned long long *accepted;
    // work list of this colour: (y, x, real, -).  real == 0 marks a VIRTUAL
    // lattice position (outside the cube or masked) whose window intersects the
    // cube: the deferred scheme visits it only to apply the pending updates, so
    // that the windows of one launch tile the cube exactly.
    const int4 *spx;
    double min_b[3], max_b[3], amp[3];
    double ra;
    uint64_t seed;
    // tile origin inside the global cube (multi-GPU tiling): random numbers
    // are keyed by the GLOBAL spaxel index (y+gy0)*Wg + (x+gx0).
    int gy0, gx0, Wg;
    // DOMAIN of the launch: the cells [dy0,dy1) x [dx0,dx1) the windows of the part
    // being updated may touch (the part's rectangle grown by the FSF half widths,
    // clipped to the cube; the whole cube for an unpartitioned context).  The deferred
    // kernels clip every window to it, so that pending layers of a part never reach
    // cells another part (or another GPU's tile) is responsible for.
    int dy0, dy1, dx0, dx1;
    // deferred write-back state
    const uint8_t *mask;  // [H*W], 1 = spaxel is iterated
    const double *Gprev;  // [slots][Dp] pending updates of colour prev_colour
    double *Gcur;         // [slots][Dp] this launch's updates
    // k_mh_ws: up to MH_LAYERS pending colours, oldest first (see "several pending
    // layers" below); prev_cy / prev_cx / Gprev above describe the only layer for
    // the kernels that keep a single one
    int n_lay, write_back;
    // Zig-zag: every other colour class of a part walks its window positions from the
    // last to the first (k_mh, k_mh_defer, k_mh_ws: this launch's direction; k_mh_flow,
    // k_mh_pair: zig-zag enabled, the direction follows the item's colour ordinal).  A
    // colour's window is its predecessor's shifted by one column: read backwards, it meets
    // the lines the predecessor touched LAST first, still in the Infinity Cache when the
    // cube's working set exceeds it (300x300x256: 96.8 -> 83.7 us per launch).  The window
    // sums are then accumulated in that order, in every kernel alike.
    int rev;
    int lay_cy[3], lay_cx[3];
    const double *lay_G[3];
    int prev_cy, prev_cx; // colour class of the pending updates, -1 = none
    int slots_x;          // slot(y,x) = (y/fh)*slots_x + x/fw
    // external-lines mode (d3d_mh_colour_lines: a python LineModel evaluated on
    // the host): per workgroup i the spaxel ext_idx[i], its current amplitude
    // ext_in[i*3+0] (1 when the model has no Gibbs amplitude), an out-of-bounds
    // flag ext_in[i*3+1], log(u) of the acceptance test ext_in[i*3+2], and the
    // unit lines ext_lines[(i*2+0)*D ..] (current) / [(i*2+1)*D ..] (proposed).
    // Results {accepted, amplitude, delta} go to ext_out[i*3 ..].
    const int *ext_idx;
    const double *ext_in;
    const double *ext_lines;
    double *ext_out;
    int ext_gibbs;  // 1: draw the amplitude (lib/run.py:456-519), 0: model without Gibbs
    // probe mode (d3d_window_stats): evaluate probe_p at spaxel probe_sp, write
    // 5 doubles to probe_out, modify nothing.
    int probe;
    int probe_sp;
    double probe_p[3];
    double *probe_out;
    // Proposals of the whole sweep, [H*W] by local spaxel index, computed by k_mh_proposals
    // before the sweep's first colour (or NULL: every update makes its own).  A proposal
    // depends on the spaxel's parameters at the START of the sweep and on its Philox stream
    // only (lib/run.py:369-388), so the three tan, the log and the two Philox blocks need
    // not sit on the critical path of a small colour launch (its prepare wavefront).
    const MHProposal *props;
    // Z-blocked sweeps of deep cubes (k_mh_ws<..., ZBK>, k_mh_zdecide): the spectrum is cut into
    // blocks of z_db channels, one workgroup per (window, block); z_part takes the blocks' wave
    // sums [item][block][wave][8], z_E the LSF-convolved unit lines of the update,
    // [2][slots][Dp] (current, proposed), from which k_mh_zdecide forms the G row.
    double *z_part;
    double *z_E;
    int z_nb, z_db, z_slots;
    // Batched chains (k_mh_ws<..., BATCH>, d3d_mh_sweeps_batch): R chains of one geometry in ONE
    // launch per colour class, grid = R x b_items; `batch` holds what differs between them, the
    // G buffers by index (all chains rotate theirs alike).
    const struct MHChainArgs *batch;
    int b_items, b_gcur, b_lay_g[3];
    // Staggered completion (EXPERIMENTS builds, option mh_prio; mh_stagger): half of a
    // colour's windows finish streaming before the other half, so that their decisions
    // overlap the others' streams (measured flat: DESIGN.md section 3).
    int prio;
    // k_mh_small (round 4, csrc/d3d_mh_small.h): the sweep's LINE TABLE, [H*W][2][Dp] -- the
    // LSF-convolved unit lines of every spaxel's current and proposed (c, w), built with the
    // proposals by k_mh_line_table before the sweep's first colour -- and ceil(2^16 / fw), for
    // position -> (row, column) of the window without a division.
    const double *ltab;
    int fw_inv;
    // ... and the relative position tables of the context (d3d_mh_small.h: MHPos), with the
    // row of this launch's pair of colour classes
    const double *ptab;
    int ptab_row[2];  // (one per pending layer the kernel can apply)
#ifdef D3D_EXPERIMENTS
    // k_mh_ws phase stamps (100 MHz wall clock), 8 slots per workgroup: 0 entry,
    // 1 setup done, 2 window streamed, 3 prepare wavefront done, 4 update written
    // (tools/mh_phases.py)
    unsigned long long *stamp;
#endif
};

#ifdef D3D_EXPERIMENTS
#define D3D_MH_STAMP(at, k, who)                                                      \
    do {                                                                              \
        if (P.stamp && threadIdx.x == (who)) P.stamp[(long)(at) * 8 + (k)] = wall_clock64(); \
    } while (0)
#else
#define D3D_MH_STAMP(at, k, who)
#endif

constexpr int MH_LAYERS = 3;  // pending colours k_mh_ws can apply in one pass
constexpr int MH_WS_MAX_DP = 512;  // deepest cube k_mh_ws takes (512 streaming threads, thread <-> channel)

// Staggered completion (MHArgs::prio; measured flat): 1..15 -- the
// workgroups with bit prio-1 of their index set run at raised wave priority; 16 + n -- the
// odd workgroups start n x 0.8 us late.
// EXPERIMENTS builds only: even the untaken test costs the default kernel 0.7 us per launch
// (the launch's first instructions wait for one more kernel argument).
__device__ __forceinline__ void mh_stagger(int prio) {
    prio &= 63;  // (bits 6, 7: timing-only switches of mh_ws_run)
    if (prio <= 0) return;
    if (prio < 16) {
        if ((blockIdx.x >> (prio - 1)) & 1) __builtin_amdgcn_s_setprio(2);
    } else if (blockIdx.x & 1) {
        for (int i = 16; i < prio; ++i) __builtin_amdgcn_s_sleep(30);
    }
}

__host__ __device__ inline size_t mh_lds_doubles(int NT, int HL, int Dp, int N, int npos,
                                                 int M = 1) {
    const int G = NT / HL;
    // taps | position table (1 + M ints per position, in (1+M) doubles) | 4 pending
    // G rows per layer | group partial sums | two unit lines | G | block sums
    return (size_t)npos + (size_t)(1 + M) * npos + 4 * (size_t)M * Dp + (size_t)G * 3 * Dp +
           2 * (size_t)N + Dp + 8 * (NT / 64) + 8;
}

struct MHShared {
    double *fsf, *gp, *red, *gO, *gN, *G, *sum;
    int *pos;
};

__device__ __forceinline__ MHShared mh_carve(double *smem, int NT, int HL, int Dp, int N,
                                              int npos, int M = 1) {
    MHShared S;
    const int G = NT / HL;
    S.fsf = smem;
    S.pos = reinterpret_cast<int *>(S.fsf + npos);  // 1 + M ints per position
    S.gp = S.fsf + (size_t)(2 + M) * npos;
    S.red = S.gp + 4 * (size_t)M * Dp;
    S.gO = S.red + (size_t)G * 3 * Dp;
    S.gN = S.gO + N;
    S.G = S.gN + N;
    S.sum = S.G + Dp;
    return S;
}

// Residual update coefficient of one channel: err += f * G, G = a_old*E_old -
// r*E_end (lib/run.py:508-515).  One definition (explicit fma) so that the
// owner's kernel and a neighbour tile's replay (k_apply_updates) round alike.
__device__ __forceinline__ double residual_coeff(double a_old, double EO, double r, double Eend) {
    return fma(-r, Eend, a_old * EO);
}

// ---- the decision, in three steps shared by every MH kernel ---------------

struct MHProposal {
    double a_old, c_old, w_old;
    double pn[3];
    double log_u;  // log of the acceptance uniform
    bool oob;
    uint32_t gsp;  // global spaxel index (Philox key)
};

// lib/run.py:369-388: Cauchy jump from the given current parameters and bounds test
// (the arithmetic of mh_propose; k_mh_chain calls it with parameters it keeps in LDS).
__device__ __forceinline__ MHProposal mh_propose_from(const MHArgs &P, double a_old, double c_old,
                                                      double w_old, uint32_t gsp, uint32_t sweep) {
    MHProposal q;
    q.gsp = gsp;
    q.a_old = a_old;
    q.c_old = c_old;
    q.w_old = w_old;
    double u_acc = 0.5;
    if (P.probe) {
        q.pn[0] = P.probe_p[0];
        q.pn[1] = P.probe_p[1];
        q.pn[2] = P.probe_p[2];
    } else {
        // lib/run.py:570-579: p + amp * tan(U(-pi/2, pi/2))
        const U2 u0 = philox_pair(P.seed, q.gsp, sweep, BLK_JUMP_AC);
        const U2 u1 = philox_pair(P.seed, q.gsp, sweep, BLK_JUMP_W);
        const double PI = 3.141592653589793;
        q.pn[0] = q.a_old + P.amp[0] * tan(PI * (u0.x - 0.5));
        q.pn[1] = q.c_old + P.amp[1] * tan(PI * (u0.y - 0.5));
        q.pn[2] = q.w_old + P.amp[2] * tan(PI * (u1.x - 0.5));
        u_acc = u1.y;
    }
    q.log_u = log(u_acc);
    // lib/run.py:379-384
    q.oob = false;
#pragma unroll
    for (int k = 0; k < 3; ++k)
        q.oob = q.oob || (q.pn[k] < P.min_b[k]) || (q.pn[k] > P.max_b[k]);
    return q;
}

// lib/run.py:369-388: Cauchy jump from the current parameters and bounds test.
// Every calling thread computes the same numbers.
__device__ __forceinline__ MHProposal mh_propose(const MHArgs &P, int sp, uint32_t sweep) {
    MHProposal q;
    const int ly = sp / P.W, lx = sp - ly * P.W;
    q.gsp = (uint32_t)((ly + P.gy0) * P.Wg + (lx + P.gx0));
    if (P.ext_lines) {
        // proposal made on the host: only the amplitude, the bounds verdict and
        // log(u) come in; the lines themselves are read by the caller
        const double *in3 = P.ext_in + (long)blockIdx.x * 3;
        q.a_old = in3[0];
        q.c_old = q.w_old = 0.0;
        q.pn[0] = q.a_old;
        q.pn[1] = q.pn[2] = 0.0;
        q.oob = in3[1] != 0.0;
        q.log_u = in3[2];
        return q;
    }
    return mh_propose_from(P, P.params[(long)sp * 3 + 0], P.params[(long)sp * 3 + 1],
                           P.params[(long)sp * 3 + 2], q.gsp, sweep);
}

// The proposal of an update: taken from the sweep's table when there is one (same function,
// same arguments, same bits as made here).
__device__ __forceinline__ MHProposal mh_proposal_of(const MHArgs &P, int sp, uint32_t sweep) {
    if (P.props && !P.ext_lines && !P.probe) return P.props[sp];
    return mh_propose(P, sp, sweep);
}

// LSF-convolved unit lines of channel ch from the zero-extended unit lines in
// gO / gN (closed form of convolve_1d, lib/convolution.py:89-120).
__device__ __forceinline__ void mh_lsf(const MHArgs &P, const double *gO, const double *gN, int ch,
                                       double *EO, double *EN) {
    double eo = 0.0, en = 0.0;
    if (ch < P.D) {
        if (P.ntaps > 0) {
            for (int t = 0; t < P.ntaps; ++t) {
                const int j = (ch + P.shift[t]) & (P.N - 1);
                const double wt = P.weight[t];
                eo = fma(wt, gO[j], eo);
                en = fma(wt, gN[j], en);
            }
        } else {
            eo = gO[ch];
            en = gN[ch];
        }
    }
    *EO = eo;
    *EN = en;
}

// From the per-channel window sums to the new state, in three steps (mh_finish strings
// them together; k_mh_chain gives the decision to a wavefront of its own):
//   mh_channel_sums   thread <-> channel: the seven sums over the channels of one
//                     wavefront, into S.sum[wave][0..6]
//   mh_decide_wave    ONE wavefront (every lane the same numbers): totals, accept,
//                     Gibbs draw; state written, verdict left in LDS
//   mh_update_coeff   thread <-> channel: the residual update coefficient G[z]
// (the seven sums of this thread's wavefront, valid in lane 63; `red`: the group partial sums)
__device__ __forceinline__ void mh_channel_sums_regs(const MHArgs &P, const double *red,
                                                     const MHProposal &q, int ch, int G, double EO,
                                                     double EN, double (&sums)[7]) {
    const int Dp = P.Dp, D = P.D;
    const double a_new = q.pn[0];  // the proposal keeps the amplitude (amp[0] = 0 with Gibbs)
    const double Lo = q.a_old * EO;
    double Az = 0.0, Bz = 0.0, Cz = 0.0;
    if (ch < D) {
#pragma unroll 4
        for (int gg = 0; gg < G; ++gg) {
            const double *r = red + (size_t)gg * 3 * Dp + ch;
            Az += r[0];
            Bz += r[Dp];
            Cz += r[2 * Dp];
        }
    }
    const double d = Lo - a_new * EN;  // old minus new contribution per unit f
    const double ulB = Az + Lo * Bz;   // sum_pos f v ul
    sums[0] = d * Az;
    sums[1] = d * d * Bz;
    sums[2] = Cz;
    sums[3] = EO * EO * Bz;
    sums[4] = EO * ulB;
    sums[5] = EN * EN * Bz;
    sums[6] = EN * ulB;
#pragma unroll
    for (int k = 0; k < 7; ++k) sums[k] = wave_sum_dpp63(sums[k]);
}

__device__ __forceinline__ void mh_channel_sums(const MHArgs &P, const MHShared &S,
                                                const MHProposal &q, int ch, int G, double EO,
                                                double EN, int first) {
    double sums[7];
    mh_channel_sums_regs(P, S.red, q, ch, G, EO, EN, sums);
    const int wave = (threadIdx.x >> 6) - first;
    if ((threadIdx.x & 63) == 63) {
#pragma unroll
        for (int k = 0; k < 7; ++k) S.sum[wave * 8 + k] = sums[k];
    }
}

// The decision proper, from the seven totals (every lane of the calling wavefront the same
// numbers): accept, Gibbs draw; the lanes with `writes` set store the new state.
__device__ __forceinline__ void mh_decide_core(const MHArgs &P, const MHProposal &q, int sp,
                                               uint32_t sweep, const double (&tot)[7],
                                               const U2 &u_gibbs, bool writes, bool *accept_out,
                                               double *r_out) {
    const double delta = -tot[0] - 0.5 * tot[1];  // ar_old - ar_new, lib/run.py:426
    // ---- MH accept (lib/run.py:435-445) --------------------------------
    const bool accept = (q.log_u < delta) && !q.oob;
    // after an accepted move err = ul - a_new*f*E_new, ul is unchanged
    const double s_ee = accept ? tot[5] : tot[3];
    const double s_eu = accept ? tot[6] : tot[4];
    // ---- Gibbs draw of the amplitude (lib/run.py:456-499) --------------
    double r;
    if (P.ext_lines && !P.ext_gibbs) {
        r = q.a_old;  // model without a Gibbs amplitude: the lines are absolute
    } else {
        const double ro = P.ra / (1.0 + P.ra * s_ee);
        const double mu = ro * s_eu;
        uint32_t blk = BLK_GIBBS;
        r = truncated_normal<true>(P.min_b[0], P.max_b[0], mu, sqrt(ro), u_gibbs, P.seed, q.gsp,
                                   sweep, &blk);
    }
    *accept_out = accept;
    *r_out = r;
    if (writes) {
        if (P.ext_lines) {
            double *o3 = P.ext_out + (long)blockIdx.x * 3;
            o3[0] = accept ? 1.0 : 0.0;
            o3[1] = r;
            o3[2] = delta;
        } else {
            if (P.prev) {  // remembered for d3d_export_updates (tiled multi-GPU replay)
                P.prev[(long)sp * 3 + 0] = q.a_old;
                P.prev[(long)sp * 3 + 1] = q.c_old;
                P.prev[(long)sp * 3 + 2] = q.w_old;
            }
            P.params[(long)sp * 3 + 0] = r;
            P.params[(long)sp * 3 + 1] = accept ? q.pn[1] : q.c_old;
            P.params[(long)sp * 3 + 2] = accept ? q.pn[2] : q.w_old;
        }
        P.dlog[sp] = delta;
        if (accept) atomicAdd(P.accepted, 1ULL);
    }
}

// One wavefront takes the decision (every lane the same numbers) and leaves
// {accepted, amplitude} in the spare slots behind the wave sums: the fp64
// special functions of the truncated normal would otherwise be issued by
// every wavefront of every resident workgroup at the same moment.
__device__ __forceinline__ void mh_decide_wave(const MHArgs &P, const MHShared &S,
                                               const MHProposal &q, int sp, uint32_t sweep, int nw,
                                               const U2 &u_gibbs) {
    double *verdict = S.sum + 8 * nw;
    double tot[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        double t = 0.0;
        for (int wv = 0; wv < nw; ++wv) t += S.sum[wv * 8 + k];
        tot[k] = t;
    }
    const double ar_old = 0.5 * tot[2];
    const double delta = -tot[0] - 0.5 * tot[1];  // ar_old - ar_new, lib/run.py:426
    const bool lead = (threadIdx.x & 63) == 0;
    if (P.probe) {
        if (lead) {
            P.probe_out[0] = ar_old;
            P.probe_out[1] = ar_old - delta;
            P.probe_out[2] = delta;
            P.probe_out[3] = tot[3];
            P.probe_out[4] = tot[4];
        }
        return;
    }
    bool accept;
    double r;
    mh_decide_core(P, q, sp, sweep, tot, u_gibbs, lead, &accept, &r);
    if (lead) {
        verdict[0] = accept ? 1.0 : 0.0;
        verdict[1] = r;
    }
}

// err_final = ul - f*E_end*r = e + f*(a_old*E_old - r*E_end)  (lib/run.py:508-515)
__device__ __forceinline__ double mh_update_coeff(const MHArgs &P, const MHShared &S,
                                                  const MHProposal &q, int ch, double EO, double EN,
                                                  int nw) {
    const double *verdict = S.sum + 8 * nw;
    const bool accept = verdict[0] != 0.0;
    const double r = verdict[1];
    return (ch < P.D) ? residual_coeff(q.a_old, EO, r, accept ? EN : EO) : 0.0;
}

// ch = this thread's channel (threads with ch >= D carry zeros), `first` = index of the
// first of the nw wavefronts that call this (they are consecutive).  Contains two block
// barriers that every thread of the workgroup must reach.  Returns false in probe mode
// and to non-callers.
__device__ __forceinline__ bool mh_finish(const MHArgs &P, const MHShared &S, const MHProposal &q,
                                          int sp, uint32_t sweep, int ch, int G, double EO,
                                          double EN, int first, int nw, bool caller,
                                          double *Gz_out, const U2 *u_pre = nullptr,
                                          long stamp_at = -1) {
    // the uniforms of the Gibbs draw depend on nothing the window pass produces:
    // drawn here, ahead of the barrier, they are off the critical tail
    U2 u_gibbs = {0.5, 0.5};
    if (caller) {
        u_gibbs = u_pre ? *u_pre : philox_pair(P.seed, q.gsp, sweep, BLK_GIBBS);
        mh_channel_sums(P, S, q, ch, G, EO, EN, first);
    }
    __syncthreads();
    if (stamp_at >= 0) D3D_MH_STAMP(stamp_at, 6, 0);  // channel sums in LDS
    if (caller && (int)(threadIdx.x >> 6) == first) mh_decide_wave(P, S, q, sp, sweep, nw, u_gibbs);
    __syncthreads();
    if (stamp_at >= 0) D3D_MH_STAMP(stamp_at, 7, 0);  // verdict in LDS
    if (!caller || P.probe) {
        *Gz_out = 0.0;
        return false;
    }
    *Gz_out = mh_update_coeff(P, S, q, ch, EO, EN, nw);
    return true;
}

// The whole decision with every thread of an NT-thread block taking part
// (thread t <-> channel t); group partial sums must be in S.red (no barrier
// needed before the call).
template <int NT>
__device__ __forceinline__ bool mh_decide(const MHArgs &P, const MHShared &S, int sp,
                                          uint32_t sweep, double *Gz_out) {
    const int tid = threadIdx.x;
    const MHProposal q = mh_proposal_of(P, sp, sweep);
    if (tid < P.N) {
        if (P.ext_lines) {
            const double *L = P.ext_lines + (long)blockIdx.x * 2 * P.D;
            S.gO[tid] = (tid < P.D) ? L[tid] : 0.0;
            S.gN[tid] = (tid < P.D) ? L[P.D + tid] : 0.0;
        } else {
            S.gO[tid] = (tid < P.D) ? unit_gaussian((double)tid, q.c_old, q.w_old) : 0.0;
            S.gN[tid] = (tid < P.D) ? unit_gaussian((double)tid, q.pn[1], q.pn[2]) : 0.0;
        }
    }
    __syncthreads();
    double EO, EN;
    mh_lsf(P, S.gO, S.gN, tid, &EO, &EN);
    return mh_finish(P, S, q, sp, sweep, tid, NT / P.HL, EO, EN, 0, NT / 64, true, Gz_out);
}

#define D3D_ACCUM(e, v, f)                  \
    do {                                    \
        const double fvx_ = (f) * (v).x;    \
        const double fvy_ = (f) * (v).y;    \
        sA.x = fma(fvx_, (e).x, sA.x);      \
        sA.y = fma(fvy_, (e).y, sA.y);      \
        sB.x = fma((f), fvx_, sB.x);        \
        sB.y = fma((f), fvy_, sB.y);        \
        sC.x = fma((v).x * (e).x, (e).x, sC.x); \
        sC.y = fma((v).y * (e).y, (e).y, sC.y); \
    } while (0)

// Immediate write-back.  MAXIT > 0: the err window stays in MAXIT double2
// registers per thread between the passes; MAXIT == 0: pass 2 re-reads it.
template <int NT, int MAXIT>
__global__ __launch_bounds__(NT) void k_mh(MHArgs P, uint32_t sweep) {
    extern __shared__ double smem[];
    const int tid = threadIdx.x;
    const int HL = P.HL, Dp = P.Dp;
    const int G = NT / HL;
    const int g = tid / HL, zl = tid - g * HL;
    const bool active = g < G;
    const int fhh = (P.fh - 1) / 2, fhw = (P.fw - 1) / 2;
    const MHShared S = mh_carve(smem, NT, HL, Dp, P.N, P.npos);

    int sp;
    if (P.probe) {
        sp = P.probe_sp;
    } else if (P.ext_lines) {
        sp = P.ext_idx[blockIdx.x];
    } else {
        const int4 ent = P.spx[blockIdx.x];
        sp = ent.x * P.W + ent.y;
    }
    const int y = sp / P.W, x = sp - y * P.W;

    for (int p = tid; p < P.npos; p += NT) S.fsf[p] = P.fsf[p];
    __syncthreads();

    // ---- pass 1: window sums ------------------------------------------------
    constexpr int NREG = MAXIT > 0 ? MAXIT : 1;
    double2 ereg[NREG];
    double2 sA = make_double2(0.0, 0.0), sB = sA, sC = sA;
    if constexpr (MAXIT > 0) {
#pragma unroll
        for (int it = 0; it < NREG; ++it) {
            const int pw = g + it * G;
            const int p = P.rev ? P.npos - 1 - pw : pw;
            const int dy = p / P.fw, dx = p - dy * P.fw;
            const int yy = y + dy - fhh, xx = x + dx - fhw;
            const bool ok = active && pw < P.npos && yy >= 0 && yy < P.H && xx >= 0 && xx < P.W;
            double2 e = make_double2(0.0, 0.0), v = e;
            double f = 0.0;
            if (ok) {
                const long idx = ((long)yy * P.W + xx) * Dp + 2 * zl;
                e = *reinterpret_cast<const double2 *>(P.err + idx);
                v = *reinterpret_cast<const double2 *>(P.ivar + idx);
                f = S.fsf[p];
            }
            ereg[it] = e;
            D3D_ACCUM(e, v, f);
        }
    } else {
        if (active) {
            for (int pw = g; pw < P.npos; pw += G) {
                const int p = P.rev ? P.npos - 1 - pw : pw;
                const int dy = p / P.fw, dx = p - dy * P.fw;
                const int yy = y + dy - fhh, xx = x + dx - fhw;
                if (yy < 0 || yy >= P.H || xx < 0 || xx >= P.W) continue;
                const long idx = ((long)yy * P.W + xx) * Dp + 2 * zl;
                const double2 e = *reinterpret_cast<const double2 *>(P.err + idx);
                const double2 v = *reinterpret_cast<const double2 *>(P.ivar + idx);
                const double f = S.fsf[p];
                D3D_ACCUM(e, v, f);
            }
        }
    }
    if (active) {
        double *r = S.red + (size_t)g * 3 * Dp + 2 * zl;
        r[0] = sA.x;
        r[1] = sA.y;
        r[Dp] = sB.x;
        r[Dp + 1] = sB.y;
        r[2 * Dp] = sC.x;
        r[2 * Dp + 1] = sC.y;
    }

    double Gt;
    if (!mh_decide<NT>(P, S, sp, sweep, &Gt)) return;
    if (tid < Dp) S.G[tid] = Gt;
    __syncthreads();

    // ---- pass 2: write the window back ------------------------------------
    if (!active) return;
    const double2 Gz = *reinterpret_cast<const double2 *>(S.G + 2 * zl);
    if constexpr (MAXIT > 0) {
#pragma unroll
        for (int it = 0; it < NREG; ++it) {
            const int pw = g + it * G;
            const int p = P.rev ? P.npos - 1 - pw : pw;
            const int dy = p / P.fw, dx = p - dy * P.fw;
            const int yy = y + dy - fhh, xx = x + dx - fhw;
            const bool ok = pw < P.npos && yy >= 0 && yy < P.H && xx >= 0 && xx < P.W;
            if (ok) {
                const long idx = ((long)yy * P.W + xx) * Dp + 2 * zl;
                const double f = S.fsf[p];
                double2 e = ereg[it];
                e.x = fma(f, Gz.x, e.x);
                e.y = fma(f, Gz.y, e.y);
                *reinterpret_cast<double2 *>(P.err + idx) = e;
            }
        }
    } else {
        for (int pw = g; pw < P.npos; pw += G) {
            const int p = P.rev ? P.npos - 1 - pw : pw;
            const int dy = p / P.fw, dx = p - dy * P.fw;
            const int yy = y + dy - fhh, xx = x + dx - fhw;
            if (yy < 0 || yy >= P.H || xx < 0 || xx >= P.W) continue;
            const long idx = ((long)yy * P.W + xx) * Dp + 2 * zl;
            const double f = S.fsf[p];
            double2 e = *reinterpret_cast<const double2 *>(P.err + idx);
            e.x = fma(f, Gz.x, e.x);
            e.y = fma(f, Gz.y, e.y);
            *reinterpret_cast<double2 *>(P.err + idx) = e;
        }
    }
}

// Spaxel of colour class (cy,cx) whose window covers coordinate q along one
// axis (period per, half width hw), or -1 when that spaxel lies outside [0,n).
__device__ __forceinline__ int covering_coord(int q, int c, int per, int hw, int n) {
    int m = (q - c) % per;
    if (m < 0) m += per;
    int s = q - m;            // largest coordinate <= q of the class
    if (q - s > hw) s += per;  // nearer one is above
    return (s >= 0 && s < n) ? s : -1;
}

// Deferred write-back (see the section header).  Position table in LDS, per
// window position p: [0] local spaxel index of the voxel column (-1 = outside
// the cube), [1] tap index of the pending update there (-1 = none), [2] which
// of the <= 4 staged pending G rows.
template <int NT>
__global__ __launch_bounds__(NT) void k_mh_defer(MHArgs P, uint32_t sweep) {
    extern __shared__ double smem[];
    const int tid = threadIdx.x;
    const int HL = P.HL, Dp = P.Dp;
    const int G = NT / HL;
    const int g = tid / HL, zl = tid - g * HL;
    const bool active = g < G;
    const int fhh = (P.fh - 1) / 2, fhw = (P.fw - 1) / 2;
    const MHShared S = mh_carve(smem, NT, HL, Dp, P.N, P.npos);

    const int4 ent = P.spx[blockIdx.x];
    const int y = ent.x, x = ent.y;  // may lie outside the cube when virtual
    const bool real = ent.z != 0;
    const int sp = y * P.W + x;
    if (!real && P.prev_cy < 0) return;  // nothing pending, nothing to do

    // the <= 2 x 2 pending spaxels that cover this window
    int psy[2], psx[2];
    if (P.prev_cy >= 0) {
        psy[0] = covering_coord(max(y - fhh, P.dy0), P.prev_cy, P.fh, fhh, P.H);
        psy[1] = covering_coord(min(y + fhh, P.dy1 - 1), P.prev_cy, P.fh, fhh, P.H);
        psx[0] = covering_coord(max(x - fhw, P.dx0), P.prev_cx, P.fw, fhw, P.W);
        psx[1] = covering_coord(min(x + fhw, P.dx1 - 1), P.prev_cx, P.fw, fhw, P.W);
    } else {
        psy[0] = psy[1] = psx[0] = psx[1] = -1;
    }
    for (int p = tid; p < P.npos; p += NT) {
        S.fsf[p] = P.fsf[p];
        const int dy = p / P.fw, dx = p - dy * P.fw;
        const int yy = y + dy - fhh, xx = x + dx - fhw;
        int vox = -1, tap = -1, sel = 0;
        if (yy >= P.dy0 && yy < P.dy1 && xx >= P.dx0 && xx < P.dx1) {
            vox = yy * P.W + xx;
            if (P.prev_cy >= 0) {
                const int sy = covering_coord(yy, P.prev_cy, P.fh, fhh, P.H);
                const int sx = covering_coord(xx, P.prev_cx, P.fw, fhw, P.W);
                if (sy >= 0 && sx >= 0 && P.mask[sy * P.W + sx]) {
                    tap = (yy - sy + fhh) * P.fw + (xx - sx + fhw);
                    sel = (sy == psy[0] ? 0 : 2) + (sx == psx[0] ? 0 : 1);
                }
            }
        }
        S.pos[3 * p + 0] = vox;
        S.pos[3 * p + 1] = tap;
        S.pos[3 * p + 2] = sel;
    }
    // stage the pending G rows (zeros where there is none)
    for (int i = tid; i < 4 * Dp; i += NT) {
        const int q = i / Dp, z = i - q * Dp;
        const int sy = psy[q >> 1], sx = psx[q & 1];
        double gv = 0.0;
        if (sy >= 0 && sx >= 0 && P.mask[sy * P.W + sx])
            gv = P.Gprev[((long)(sy / P.fh) * P.slots_x + sx / P.fw) * Dp + z];
        S.gp[i] = gv;
    }
    __syncthreads();

    // ---- the single pass: apply the pending update, write, accumulate ------
    double2 sA = make_double2(0.0, 0.0), sB = sA, sC = sA;
    if (active) {
#pragma unroll 4
        for (int pw = g; pw < P.npos; pw += G) {
            const int p = P.rev ? P.npos - 1 - pw : pw;
            const int vox = S.pos[3 * p + 0];
            if (vox < 0) continue;
            const int tap = S.pos[3 * p + 1];
            const long idx = (long)vox * Dp + 2 * zl;
            double2 e = *reinterpret_cast<const double2 *>(P.err + idx);
            const double2 v = *reinterpret_cast<const double2 *>(P.ivar + idx);
            if (tap >= 0) {
                const double fp = S.fsf[tap];
                const double2 gz =
                    *reinterpret_cast<const double2 *>(S.gp + S.pos[3 * p + 2] * Dp + 2 * zl);
                e.x = fma(fp, gz.x, e.x);
                e.y = fma(fp, gz.y, e.y);
                *reinterpret_cast<double2 *>(P.err + idx) = e;
            }
            const double f = S.fsf[p];
            D3D_ACCUM(e, v, f);
        }
        if (!real) return;
        double *r = S.red + (size_t)g * 3 * Dp + 2 * zl;
        r[0] = sA.x;
        r[1] = sA.y;
        r[Dp] = sB.x;
        r[Dp + 1] = sB.y;
        r[2 * Dp] = sC.x;
        r[2 * Dp + 1] = sC.y;
    }

    if (!real) return;
    double Gt;
    if (!mh_decide<NT>(P, S, sp, sweep, &Gt)) return;
    if (tid < Dp) P.Gcur[((long)(y / P.fh) * P.slots_x + x / P.fw) * Dp + tid] = Gt;
}

// Wave-specialised deferred kernel: NS streaming threads run the window pass
// while ONE extra wavefront computes everything of the decision that does not
// depend on the window -- proposal (Philox, tan), both unit lines (exp) and
// their LSF convolution, channel by channel into LDS -- so that only the short
// tail (sums -> accept -> truncated normal; streaming thread t <-> channel t)
// follows the pass.  The prepare wavefront works in its own LDS region (no
// block barrier while the others stream) and is done long before the stream
// (tools/mh_phases.py).  320 threads and ~29 KB of LDS per workgroup at D = 128:
// four workgroups per CU, so that every window of a colour launch is resident
// at once.  Same arithmetic, same summation order, bit-identical results as
// k_mh_defer.
__host__ __device__ inline size_t mh_ws_lds_doubles(int NS, int HL, int Dp, int N, int npos,
                                                    int M = 1) {
    return mh_lds_doubles(NS, HL, Dp, N, npos, M) + (size_t)Dp + 16;
}

// UV = true: the 1/variance cube is one constant (the reference's default when
// no variance is given, lib/run.py:171-178, and no NaN voxel): the streaming
// threads take it from P.ivar_uniform instead of reading SLOT_IVAR -- 16 instead
// of 24 bytes per window voxel, same arithmetic, bit-identical results.
//
// The workgroup's work on one window, in steps shared by k_mh_ws (one launch per
// colour) and k_mh_flow (one launch per sweep):
//   mh_ws_preds    the <= 2 x 2 spaxels of every pending colour that cover the window
//   mh_ws_table    position table + taps into LDS (geometry only)
//   mh_ws_gp_*     their G rows into LDS
//   mh_ws_run      window pass + prepare wavefront + decision + G row out
//
// Several pending layers (template capacity M, I.n_lay <= M in use).  Writing the
// residual costs about twice what reading it does on this part (the uniform-
// variance variant, half reads and half writes, streams at 4.6 TB/s where the
// general one reaches 5.7).  So the pass need not write e + f G back after every
// colour: up to M colours stay pending as (colour, G rows) layers, a launch
// applies them in registers, oldest first -- the same fma sequence as if each had
// been written back in turn, so the chain stays bit-identical -- and only the
// launch that finds M layers pending stores the result and starts over.  One
// residual write per M colours instead of one per colour.
// Z-blocked form (ZBK): what a workgroup needs to know about its block beyond the argument
// copy whose counts (D, Dp, HL, N) and pointers (err, ivar, G rows) describe the block alone.
struct MHZ {
    int z0;         // first channel of the block
    int Nfull;      // padded length of the whole spectrum (power of two)
    int Dfull;      // depth of the whole spectrum
    long zs;        // doubles between two spaxels' spectra (the cube's Dp)
};

struct MHWsItem {
    int y, x, real;
    int n_lay, write_back;
    int rev;  // window positions last to first (MHArgs::rev)
    int lay_cy[MH_LAYERS], lay_cx[MH_LAYERS];  // colour class of each pending layer
    // the <= 2 x 2 spaxels of layer j that cover this window (-1 = none).  Only ever
    // indexed with compile-time constants: a dynamic index would put it in scratch.
    int psy0[MH_LAYERS], psy1[MH_LAYERS], psx0[MH_LAYERS], psx1[MH_LAYERS];
    const double *lay_G[MH_LAYERS];
    double *Gcur;
};

// Item header from the launch arguments (k_mh_ws: every workgroup the same layers).
__device__ __forceinline__ void mh_ws_layers_from_args(const MHArgs &P, MHWsItem &I) {
    I.n_lay = P.n_lay;
    I.write_back = P.write_back;
    I.rev = P.rev;
#pragma unroll
    for (int j = 0; j < MH_LAYERS; ++j) {
        I.lay_cy[j] = P.lay_cy[j];
        I.lay_cx[j] = P.lay_cx[j];
        I.lay_G[j] = P.lay_G[j];
    }
    I.Gcur = P.Gcur;
}

template <int M>
__device__ __forceinline__ void mh_ws_preds(const MHArgs &P, MHWsItem &I) {
    const int fhh = (P.fh - 1) / 2, fhw = (P.fw - 1) / 2;
    const int y = I.y, x = I.x;
#pragma unroll
    for (int j = 0; j < M; ++j) {
        if (j < I.n_lay) {
            I.psy0[j] = covering_coord(max(y - fhh, P.dy0), I.lay_cy[j], P.fh, fhh, P.H);
            I.psy1[j] = covering_coord(min(y + fhh, P.dy1 - 1), I.lay_cy[j], P.fh, fhh, P.H);
            I.psx0[j] = covering_coord(max(x - fhw, P.dx0), I.lay_cx[j], P.fw, fhw, P.W);
            I.psx1[j] = covering_coord(min(x + fhw, P.dx1 - 1), I.lay_cx[j], P.fw, fhw, P.W);
        } else {
            I.psy0[j] = I.psy1[j] = I.psx0[j] = I.psx1[j] = -1;
        }
    }
}

// needs mh_ws_preds.  Table row of position p: [0] local spaxel index of the voxel
// column (-1 = outside the launch's domain); [1+j] for layer j: tap index of that layer's
// update there | which of its <= 4 staged G rows << 16, or -1 = none.
// tap0: the caller's early load of P.fsf[threadIdx.x] (k_mh_ws issues it before the window
// prefetch, so that the table does not queue behind it), or NULL.
template <int M>
__device__ __forceinline__ void mh_ws_table(const MHArgs &P, const MHShared &S, const MHWsItem &I,
                                            int NT, const double *tap0 = nullptr) {
    constexpr int ROW = 1 + M;
    const int fhh = (P.fh - 1) / 2, fhw = (P.fw - 1) / 2;
    const int y = I.y, x = I.x;
    for (int p = threadIdx.x; p < P.npos; p += NT) {
        S.fsf[p] = (tap0 && p == (int)threadIdx.x) ? *tap0 : P.fsf[p];
        const int dy = p / P.fw, dx = p - dy * P.fw;
        const int yy = y + dy - fhh, xx = x + dx - fhw;
        const bool inside = yy >= P.dy0 && yy < P.dy1 && xx >= P.dx0 && xx < P.dx1;
        S.pos[ROW * p] = inside ? yy * P.W + xx : -1;
#pragma unroll
        for (int j = 0; j < M; ++j) {
            int code = -1;
            if (inside && j < I.n_lay) {
                const int sy = covering_coord(yy, I.lay_cy[j], P.fh, fhh, P.H);
                const int sx = covering_coord(xx, I.lay_cx[j], P.fw, fhw, P.W);
                // (a masked spaxel there left a zero G row: see mh_ws_zero_row)
                if (sy >= 0 && sx >= 0)
                    code = ((yy - sy + fhh) * P.fw + (xx - sx + fhw)) |
                           (((sy == I.psy0[j] ? 0 : 2) + (sx == I.psx0[j] ? 0 : 1)) << 16);
            }
            S.pos[ROW * p + 1 + j] = code;
        }
    }
}

// The <= 4 pending G rows of every layer into LDS (S.gp[j][q][Dp]), in two halves
// so that the loads can be in flight while the position table is computed
// (4*Dp <= K * NT values per layer, K = 4 for one or two layers, 2 for three: the
// launcher limits the layers accordingly).
// COH (k_mh_flow): the rows were written in THIS launch by workgroups of any
// XCD -- agent-scope (sc1) loads, never served from a stale L1 line.
template <int M, int K_>
struct MHGpRegs {
    static constexpr int K = K_;
    double v[M][K_];
};

template <int M, int K, bool COH, bool ZBK = false>
__device__ __forceinline__ void mh_ws_gp_load(const MHArgs &P, const MHWsItem &I, int NT,
                                              MHGpRegs<M, K> &R, long zs = 0) {
    const int Dp = P.Dp;
    // (one scalar register for all K loads: left to itself the compiler re-fetches the kernel
    // argument in front of every one of them -- four dependent scalar loads in the setup of a
    // small launch, 0.6 us per launch at 64 channels)
    int slots_x = P.slots_x;
    asm volatile("" : "+s"(slots_x));
#pragma unroll
    for (int j = 0; j < M; ++j) {
        const int py0 = I.psy0[j], py1 = I.psy1[j], px0 = I.psx0[j], px1 = I.psx1[j];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int i = threadIdx.x + k * NT;
            R.v[j][k] = 0.0;
            if (j < I.n_lay && i < 4 * Dp) {
                const int q = i / Dp, z = i - q * Dp;
                const int sy = (q >> 1) ? py1 : py0;
                const int sx = (q & 1) ? px1 : px0;
                if (sy >= 0 && sx >= 0) {
                    const double *src =
                        I.lay_G[j] + ((long)(sy / P.fh) * slots_x + sx / P.fw) * (ZBK ? zs : (long)Dp) + z;
                    if (COH)
                        R.v[j][k] = __longlong_as_double((long long)__hip_atomic_load(
                            reinterpret_cast<const unsigned long long *>(src), __ATOMIC_RELAXED,
                            __HIP_MEMORY_SCOPE_AGENT));
                    else
                        R.v[j][k] = *src;
                }
            }
        }
    }
}

template <int M, int K>
__device__ __forceinline__ void mh_ws_gp_store(const MHArgs &P, const MHShared &S, const MHWsItem &I,
                                               int NT, const MHGpRegs<M, K> &R) {
#pragma unroll
    for (int j = 0; j < M; ++j) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int i = threadIdx.x + k * NT;
            if (j < I.n_lay && i < 4 * P.Dp) S.gp[(size_t)j * 4 * P.Dp + i] = R.v[j][k];
        }
    }
}

// A virtual item that is a masked spaxel INSIDE the cube has no update: it leaves
// a zero G row, so that the next colour can apply "every lattice point inside
// the cube" without looking the mask up (one dependent load less in its setup).
template <bool COH, bool ZBK = false>
__device__ __forceinline__ void mh_ws_zero_row(const MHArgs &P, const MHWsItem &I, long zs = 0) {
    const int tid = threadIdx.x;
    if (tid < P.Dp && I.y >= 0 && I.y < P.H && I.x >= 0 && I.x < P.W) {
        double *dst =
            I.Gcur + ((long)(I.y / P.fh) * P.slots_x + I.x / P.fw) * (ZBK ? zs : (long)P.Dp) + tid;
        if (COH)
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(dst), 0ULL, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        else
            *dst = 0.0;
    }
}

// Needs S.pos / S.fsf / S.gp complete (block barrier before the call).  Contains
// block barriers only when I.real (uniform over the workgroup).
// COH (k_mh_flow): the residual and the G rows are handed from workgroup to
// workgroup inside the launch: write-through (sc1) stores and sc1 loads for
// every such byte, so that neither a release nor an acquire fence is needed
// (cdna_hip_programming.md, Guideline 16, the all-sc1 form).
// COHG: only the G row this item leaves is published coherently (k_mh_pair: the other
// colour class of the same launch reads it; the residual is handed over by nobody).
// NTV (cache policy of a context beyond the Infinity Cache): the 1/variance stream with the
// non-temporal hint, and the residual stored write-through (mh_ws_run).  When residual + 1/variance
// exceed the Infinity Cache, the read-only half should not compete for it with the half
// that the next colour class re-reads AND rewrites (300x300x256: 83.7 -> 78.9 us per
// launch, 600x600x128: 176 -> 164); when both fit, the hint costs (300x300x128: 40.7 ->
// 43.3 us).  The host chooses per context (d3d_ctx::mh_nt_ivar); same bytes, same results.
typedef double d3d_v2d __attribute__((ext_vector_type(2)));
template <bool NTV>
__device__ __forceinline__ double2 mh_load_ivar(const double *p) {
    if constexpr (NTV) {
        const d3d_v2d t = __builtin_nontemporal_load(reinterpret_cast<const d3d_v2d *>(p));
        return make_double2(t.x, t.y);
    } else {
        return *reinterpret_cast<const double2 *>(p);
    }
}

template <int U>
struct MHPre {
    int vox[U];
    double2 e[U], v[U];
};

// The loads of the first window round of a streaming thread (positions g + u G), issued
// before the workgroup's setup: S.pos is not built yet, so the voxel index is computed
// here -- the same expression as mh_ws_table's.
template <int NS, bool UV, int U, bool NTV = false, bool ZBK = false>
__device__ __forceinline__ void mh_ws_prefetch(const MHArgs &P, const MHWsItem &I, MHPre<U> &R,
                                               long zs = 0) {
    const int tid = threadIdx.x;
    const int HL = P.HL, Dp = P.Dp;
    const int G = NS / HL;
    const int g = tid / HL, zl = tid - g * HL;
    const int fhh = (P.fh - 1) / 2, fhw = (P.fw - 1) / 2;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        R.vox[u] = -1;
        R.e[u] = make_double2(0.0, 0.0);
        R.v[u] = R.e[u];
    }
    if (tid >= NS || g >= G) return;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int pw = g + u * G;
        if (pw < P.npos) {
            const int p = I.rev ? P.npos - 1 - pw : pw;
            const int dy = p / P.fw, dx = p - dy * P.fw;
            const int yy = I.y + dy - fhh, xx = I.x + dx - fhw;
            const bool inside = yy >= P.dy0 && yy < P.dy1 && xx >= P.dx0 && xx < P.dx1;
            R.vox[u] = inside ? yy * P.W + xx : -1;
        }
        const long idx = (long)max(R.vox[u], 0) * (ZBK ? zs : (long)Dp) + 2 * zl;
        R.e[u] = *reinterpret_cast<const double2 *>(P.err + idx);
#ifdef D3D_EXPERIMENTS
        if (!UV && !(P.prio & 128)) R.v[u] = mh_load_ivar<NTV>(P.ivar + idx);
#else
        if (!UV) R.v[u] = mh_load_ivar<NTV>(P.ivar + idx);
#endif
    }
}

// ZBK: the LSF-convolved unit lines of one block.  L holds the zero-extended, Nfull-periodic
// unit line on the block's channels and LSF_RL either side (L[i] <-> channel z0 - LSF_RL + i);
// the taps (all within +-LSF_RL: the host checks) in the order mh_lsf takes them.
__device__ __forceinline__ void mh_lsf_block(const MHArgs &P, const MHZ &Z, const double *LO,
                                             const double *LN, int ch, double *EO, double *EN) {
    double eo = 0.0, en = 0.0;
    if (ch < P.D) {
        if (P.ntaps > 0) {
            for (int t = 0; t < P.ntaps; ++t) {
                const int sh = P.shift[t];
                const int j = ch + LSF_RL + (sh > Z.Nfull / 2 ? sh - Z.Nfull : sh);
                const double wt = P.weight[t];
                eo = fma(wt, LO[j], eo);
                en = fma(wt, LN[j], en);
            }
        } else {
            eo = LO[ch + LSF_RL];
            en = LN[ch + LSF_RL];
        }
    }
    *EO = eo;
    *EN = en;
}

// PROPS: the launch may take its proposals from the sweep's table (MHArgs::props; the small
// launches' variants only: the others keep the table out of their code).
// ZBK: P describes ONE z-block of the window (counts and pointers; Z the rest); the run ends
// with the block's wave sums and its lines handed to k_mh_zdecide (item = the window's index
// in the launch).
template <int NS, bool UV, bool COH, int U, int M, bool COHG = COH, bool PRE = false, bool NTV = false,
          bool PROPS = false, bool ZBK = false>
__device__ __forceinline__ void mh_ws_run(const MHArgs &P, const MHShared &S, const MHWsItem &I,
                                          uint32_t sweep, long stamp_at,
                                          const MHPre<U> *pre = nullptr, const MHZ *Zp = nullptr,
                                          int item = 0, int zb = 0) {
    const long zstride = ZBK ? Zp->zs : (long)P.Dp;  // doubles between two spaxels' spectra
    constexpr int ROW = 1 + M;
    const int tid = threadIdx.x;
    const int HL = P.HL, Dp = P.Dp, N = P.N;
    const int G = NS / HL;
    // S.gO / S.gN: the prepare wavefront's zero-extended unit lines; S.G and sEN:
    // the LSF-convolved lines per channel; sq: the proposal
    double *sEN = S.sum + 8 * (NS / 64) + 8;
    MHProposal *sq = reinterpret_cast<MHProposal *>(sEN + Dp);
    static_assert(sizeof(MHProposal) <= 16 * sizeof(double), "proposal does not fit its LDS slot");
    const bool real = I.real != 0;
    const int sp = I.y * P.W + I.x;
    const bool streamer = tid < NS;
    // the uniforms of the Gibbs draw, while the first loads of the window pass fly
    U2 u_gibbs = {0.5, 0.5};
    if (!ZBK && streamer && real && !P.ext_lines)
        u_gibbs = philox_pair(P.seed, (uint32_t)((I.y + P.gy0) * P.Wg + (I.x + P.gx0)), sweep,
                              BLK_GIBBS);
    if (streamer) {
        const int g = tid / HL, zl = tid - g * HL;
        if (g < G) {
            double2 sA = make_double2(0.0, 0.0), sB = sA, sC = sA;
            // the padding channel of an odd depth carries 1/var = 0
            const double2 vu = make_double2(P.ivar_uniform,
                                            (2 * zl + 1 < P.D) ? P.ivar_uniform : 0.0);
            // raw buffer over SLOT_ERR; aux 16 = sc1.  COH (k_mh_flow / k_mh_pair, loads and
            // stores): the whole slot, which the launcher checks to be < 2 GiB.  NTV alone
            // (write-through STORES of a context beyond the Infinity Cache): a buffer of this
            // WINDOW -- base = its first cell, 32-bit offsets within (fh + 1) rows of the cube --
            // so that the policy survives residual cubes of any size (round 4: a full MUSE cube,
            // 300x300x3682, is 2.65 GB)
            typedef unsigned v4u __attribute__((ext_vector_type(4)));
            union { double2 d; v4u i; } cv;
            long rs_vox = 0;   // first cell the buffer covers
            if constexpr (NTV && !COH) {
                const int fhh_ = (P.fh - 1) / 2, fhw_ = (P.fw - 1) / 2;
                rs_vox = max(0L, (long)(I.y - fhh_) * P.W + (I.x - fhw_));
            }
            const long rs_bytes =
                (COH || NTV) ? ((long)P.H * P.W - rs_vox) * zstride * 8 - (ZBK ? Zp->z0 : 0) * 8 : 0;
            const __amdgpu_buffer_rsrc_t err_rsrc = __builtin_amdgcn_make_buffer_rsrc(
                P.err + rs_vox * zstride, 0, (int)(unsigned)min(rs_bytes, 0x7fffffffL), 0x00020000);
            // U window positions per round: all their loads are issued before the
            // first is consumed.  U = 1 when a launch fills the chip (the stream is at
            // the HBM peak; more requests in flight only add contention: 52.9 vs
            // 50.9 us per colour at 300x300x128), U = 4 for small cubes, whose few
            // workgroups are latency-bound (64^3: 14.6 -> 12.8 us per colour).
            // Positions outside the cube load voxel 0 and are skipped; the sums
            // still run in increasing p: bit-identical either way.
            // PRE: the first round's loads were issued by the kernel before the setup
            // (mh_ws_prefetch), so that the memory system is not idle while every
            // workgroup of the launch computes its table; same values, same order.
            auto issue = [&](int p0, int (&vox)[U], double2 (&e)[U], double2 (&v)[U]) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int pw = p0 + u * G;
                    const int p = I.rev ? P.npos - 1 - pw : pw;
                    vox[u] = (pw < P.npos) ? S.pos[ROW * p] : -1;
                    const long idx = (long)max(vox[u], 0) * zstride + 2 * zl;
                    if (COH) {
                        cv.i = __builtin_amdgcn_raw_buffer_load_b128(err_rsrc, (int)(idx * 8), 0, 16);
                        e[u] = cv.d;
                    } else {
                        e[u] = *reinterpret_cast<const double2 *>(P.err + idx);
                    }
                    v[u] = vu;
#ifdef D3D_EXPERIMENTS
                    // timing-only switch (wrong results): mh_prio bit 7 -- no 1/variance loads
                    if (!UV && !(P.prio & 128)) v[u] = mh_load_ivar<NTV>(P.ivar + idx);
#else
                    if (!UV) v[u] = mh_load_ivar<NTV>(P.ivar + idx);
#endif
                }
            };
            auto consume = [&](int p0, int (&vox)[U], double2 (&e)[U], double2 (&v)[U]) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (vox[u] < 0) continue;
                    const int pw = p0 + u * G;
                    const int p = I.rev ? P.npos - 1 - pw : pw;
                    const long idx = (long)vox[u] * zstride + 2 * zl;
                    // the pending layers, oldest first: e <- e + f G of each
                    bool touched = false;
#pragma unroll
                    for (int j = 0; j < M; ++j) {
                        const int code = (j < I.n_lay) ? S.pos[ROW * p + 1 + j] : -1;
                        if (code >= 0) {
                            const double fp = S.fsf[code & 0xffff];
                            const double2 gz = *reinterpret_cast<const double2 *>(
                                S.gp + ((size_t)j * 4 + (code >> 16)) * Dp + 2 * zl);
                            e[u].x = fma(fp, gz.x, e[u].x);
                            e[u].y = fma(fp, gz.y, e[u].y);
                            touched = true;
                        }
                    }
#ifdef D3D_EXPERIMENTS
                    // timing-only switch (wrong results): mh_prio bit 6 -- no residual stores
                    if (P.prio & 64) touched = false;
#endif
                    if (touched && I.write_back) {
                        // (NTV, a context beyond the Infinity Cache: write-through as well --
                        // no dirty lines left for the end of the kernel to flush; 300x300x256
                        // 79.4 -> 76.7 us per launch, nothing where everything fits)
                        if (COH || NTV) {
                            cv.d = e[u];
                            __builtin_amdgcn_raw_buffer_store_b128(
                                cv.i, err_rsrc, (int)((idx - rs_vox * zstride) * 8), 0, 16);
                        } else {
                            *reinterpret_cast<double2 *>(P.err + idx) = e[u];
                        }
                    }
                    const double f = S.fsf[p];
                    D3D_ACCUM(e[u], v[u], f);
                }
            };
            int p0 = g;
            if constexpr (PRE) {
                int vox[U];
                double2 e[U], v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    vox[u] = pre->vox[u];
                    e[u] = pre->e[u];
                    v[u] = UV ? vu : pre->v[u];
                }
                consume(p0, vox, e, v);
                p0 += U * G;
            }
            // (the deep-prefetch forms, U >= 8, hold a thread's whole window share in one or two
            // rounds: nothing to unroll)
            constexpr int UNR = U >= 8 ? 1 : 4;
#pragma unroll UNR
            for (; p0 < P.npos; p0 += U * G) {
                int vox[U];
                double2 e[U], v[U];
                issue(p0, vox, e, v);
                consume(p0, vox, e, v);
            }
            if (real) {
                double *r = S.red + (size_t)g * 3 * Dp + 2 * zl;
                r[0] = sA.x;
                r[1] = sA.y;
                r[Dp] = sB.x;
                r[Dp + 1] = sB.y;
                r[2 * Dp] = sC.x;
                r[2 * Dp + 1] = sC.y;
            }
        }
    } else if (real) {
        const int lane = tid - NS;
        const MHProposal q = PROPS ? mh_proposal_of(P, sp, sweep) : mh_propose(P, sp, sweep);
        if constexpr (ZBK) {
            // the block's channels and LSF_RL either side of the zero-extended periodic line
            // (N = block length + 2 LSF_RL here)
            for (int j = lane; j < N; j += 64) {
                int m = (Zp->z0 - LSF_RL + j) % Zp->Nfull;
                if (m < 0) m += Zp->Nfull;
                S.gO[j] = (m < Zp->Dfull) ? unit_gaussian((double)m, q.c_old, q.w_old) : 0.0;
                S.gN[j] = (m < Zp->Dfull) ? unit_gaussian((double)m, q.pn[1], q.pn[2]) : 0.0;
            }
        } else {
            for (int j = lane; j < N; j += 64) {
                S.gO[j] = (j < P.D) ? unit_gaussian((double)j, q.c_old, q.w_old) : 0.0;
                S.gN[j] = (j < P.D) ? unit_gaussian((double)j, q.pn[1], q.pn[2]) : 0.0;
            }
        }
        __builtin_amdgcn_wave_barrier();  // wave-private region: LDS is in order per wave
        for (int ch = lane; ch < Dp; ch += 64) {
            double EO, EN;
            if constexpr (ZBK) mh_lsf_block(P, *Zp, S.gO, S.gN, ch, &EO, &EN);
            else mh_lsf(P, S.gO, S.gN, ch, &EO, &EN);
            S.G[ch] = EO;
            sEN[ch] = EN;
        }
        if (lane == 0) *sq = q;
        D3D_MH_STAMP(stamp_at, 3, NS);  // prepare wavefront done (long before the stream)
    }
    D3D_MH_STAMP(stamp_at, 2, 0);
    if (!real) {
        mh_ws_zero_row<COHG, ZBK>(P, I, zstride);
        return;
    }
    __syncthreads();  // group partial sums are in S.red, the lines in S.G / sEN
    MHProposal q = {};
    double EO = 0.0, EN = 0.0;
    if (streamer) {
        q = *sq;
        if (tid < Dp) {
            EO = S.G[tid];
            EN = sEN[tid];
        }
    }
    if constexpr (ZBK) {
        // the block's share of the decision: its wave sums and its lines go to memory, the
        // totals over the blocks are k_mh_zdecide's
        if (streamer) {
            mh_channel_sums(P, S, q, tid, G, EO, EN, 0);
            if (tid < Dp) {
                const long slot = (long)(I.y / P.fh) * P.slots_x + I.x / P.fw;
                double *e0 = P.z_E + slot * zstride + Zp->z0 + tid;
                e0[0] = EO;
                e0[(long)P.z_slots * zstride] = EN;
            }
        }
        __syncthreads();
        constexpr int NWS = NS / 64;
        if (tid < NWS * 8) P.z_part[((long)item * P.z_nb + zb) * (NWS * 8) + tid] = S.sum[tid];
        return;
    }
    double Gt;
    // (Round 4, measured and dropped HERE: k_mh_small's tail -- the decision on every wavefront
    // that holds channels, one barrier instead of two -- 40.7 -> 41.5 us per launch at
    // 300x300x128, 31.4 -> 32.1 with uniform variance, same-box A/B, whether or not the other
    // wavefronts end early: with three workgroups per compute unit a second deciding wavefront
    // per window takes issue slots from the neighbours' window passes.  ONE wavefront decides.)
    if (!mh_finish(P, S, q, sp, sweep, tid, G, EO, EN, 0, NS / 64, streamer, &Gt, &u_gibbs, stamp_at))
        return;
    if (tid < Dp) {
        double *dst = I.Gcur + ((long)(I.y / P.fh) * P.slots_x + I.x / P.fw) * Dp + tid;
        if (COHG)
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(dst),
                               (unsigned long long)__double_as_longlong(Gt), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        else
            *dst = Gt;
    }
    D3D_MH_STAMP(stamp_at, 4, 0);
}

// K: registers per thread and layer that stage the pending G rows (4*Dp <= K*(NS+64))
// NL: the number of pending layers of this launch (== P.n_lay; -1: read it at run time).
// As a compile-time constant the "layer j is live" tests of the setup fold away -- and with
// them a store at a run-time index that kept the item's small arrays, and three dependent
// scratch round trips, in every workgroup's setup.
// ZBK (round 3, cubes deeper than MH_WS_MAX_DP channels): grid = windows x z-blocks of z_db
// channels; a workgroup runs the kernel on ITS block -- the arguments' counts and pointers are
// rewritten to describe the block alone, only the spaxel stride stays the cube's -- up to the
// wave sums of the decision, which k_mh_zdecide totals over the blocks.
// BATCH (round 3): the launch holds the windows of R independent chains of one geometry
// (chain-major); the chain's cubes, parameters, bounds and random stream replace the
// arguments' -- everything else (work list, taps, pending-layer geometry) is common.
template <int NS, bool UV, int U, int M, int K, int NL = -1, bool NTV = false, bool ZBK = false,
          bool BATCH = false>
__global__ __launch_bounds__(NS + 64) void k_mh_ws(MHArgs P, uint32_t sweep) {
    extern __shared__ double smem[];
    constexpr int NT = NS + 64;
    static_assert(!(ZBK && BATCH), "batched chains: cubes of one workgroup per window");
    int item = blockIdx.x;
    if constexpr (BATCH) {
        const int r = blockIdx.x / P.b_items;
        item = blockIdx.x - r * P.b_items;
        const MHChainArgs &B = P.batch[r];
        P.err = B.err;
        P.ivar = B.ivar;
        P.ivar_uniform = B.ivar_uniform;
        P.params = B.params;
        P.prev = B.prev;
        P.dlog = B.dlog;
        P.accepted = B.accepted;
        P.Gcur = B.gbuf[P.b_gcur];
#pragma unroll
        for (int j = 0; j < MH_LAYERS; ++j) P.lay_G[j] = B.gbuf[P.b_lay_g[j]];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            P.min_b[k] = B.min_b[k];
            P.max_b[k] = B.max_b[k];
            P.amp[k] = B.amp[k];
        }
        P.ra = B.ra;
        P.seed = B.seed;
    }
    if constexpr (ZBK) {
        const int item = blockIdx.x / P.z_nb, zb = blockIdx.x - item * P.z_nb;
        MHZ Z;
        Z.z0 = zb * P.z_db;
        Z.Nfull = P.N;
        Z.Dfull = P.D;
        Z.zs = P.Dp;
        const int db = min(P.z_db, P.Dp - Z.z0);  // (even: z_db and Dp are)
        P.err += Z.z0;
        P.ivar += Z.z0;
        P.Gcur += Z.z0;
#pragma unroll
        for (int j = 0; j < MH_LAYERS; ++j) P.lay_G[j] += Z.z0;
        P.D = max(0, min(db, Z.Dfull - Z.z0));
        P.Dp = db;
        P.HL = db / 2;
        P.N = db + 2 * LSF_RL;
        const MHShared S = mh_carve(smem, NS, P.HL, P.Dp, P.N, P.npos, M);
        const int4 ent = P.spx[item];
        MHWsItem I;
        I.y = ent.x;
        I.x = ent.y;
        I.real = ent.z;
        mh_ws_layers_from_args(P, I);
        if constexpr (NL >= 0) I.n_lay = NL;
        if (!I.real && (I.n_lay == 0 || !I.write_back)) {
            mh_ws_zero_row<false, true>(P, I, Z.zs);
            return;
        }
        mh_ws_preds<M>(P, I);
        MHGpRegs<M, K> gv;
        mh_ws_gp_load<M, K, false, true>(P, I, NT, gv, Z.zs);
        const double tap0 = ((int)threadIdx.x < P.npos) ? P.fsf[threadIdx.x] : 0.0;
        MHPre<U> pre;
        mh_ws_prefetch<NS, UV, U, NTV, true>(P, I, pre, Z.zs);
        mh_ws_table<M>(P, S, I, NT, &tap0);
        mh_ws_gp_store<M, K>(P, S, I, NT, gv);
        __syncthreads();
        mh_ws_run<NS, UV, false, U, M, false, true, NTV, true, true>(P, S, I, sweep, blockIdx.x, &pre, &Z,
                                                                      item, zb);
        return;
    }
    const MHShared S = mh_carve(smem, NS, P.HL, P.Dp, P.N, P.npos, M);
    D3D_MH_STAMP(blockIdx.x, 0, 0);
#ifdef D3D_EXPERIMENTS
    // where the workgroup runs: HW_ID (wave, SIMD, CU, SH, SE) | XCC_ID << 32
    if (P.stamp && threadIdx.x == 0)
        P.stamp[(long)blockIdx.x * 8 + 5] =
            (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) |
            ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 32);
    mh_stagger(P.prio);
#endif
    const int4 ent = P.spx[BATCH ? item : (int)blockIdx.x];
    MHWsItem I;
    I.y = ent.x;  // may lie outside the cube when virtual
    I.x = ent.y;
    I.real = ent.z;
    mh_ws_layers_from_args(P, I);
    if constexpr (NL >= 0) I.n_lay = NL;
    // a virtual position only matters to a launch that writes the residual back
    if (!I.real && (I.n_lay == 0 || !I.write_back)) {
        mh_ws_zero_row<false>(P, I);
        return;
    }
    // Loads in the order the setup needs them -- a wavefront's loads return in order: the
    // pending G rows and the taps first, THEN the window's first round, which flies during
    // the rest of the setup instead of holding it up.
    mh_ws_preds<M>(P, I);
    MHGpRegs<M, K> gv;
    mh_ws_gp_load<M, K, false>(P, I, NT, gv);
    const double tap0 = ((int)threadIdx.x < P.npos) ? P.fsf[threadIdx.x] : 0.0;
    MHPre<U> pre;
    mh_ws_prefetch<NS, UV, U, NTV>(P, I, pre);
    mh_ws_table<M>(P, S, I, NT, &tap0);
    mh_ws_gp_store<M, K>(P, S, I, NT, gv);
    __syncthreads();
    D3D_MH_STAMP(blockIdx.x, 1, 0);
    // (U >= 4 or the wide form: the variants of launches that do not fill the chip)
    mh_ws_run<NS, UV, false, U, M, false, true, NTV, (U >= 4 || NS != 256)>(P, S, I, sweep, blockIdx.x,
                                                                           &pre);
}

#ifdef D3D_EXPERIMENTS
// ---- whole sweeps of a small part in ONE launch: persistent workgroups ----------------------
//
// A colour launch that does not fill the chip is a latency chain: setup,
// proposal -> lines -> LSF on one wavefront, the decision, the kernel boundary -- and above
// all ONE compute unit pulling a whole window (242 KB at 128 channels, 11 x 11) through its
// own memory port at 30-45 GB/s: 12-15 us per colour however few windows the launch holds,
// 121 times per sweep.  k_mh_chain keeps ONE workgroup per lattice slot resident for all
// colours of all sweeps of the launch, and the slot's window in REGISTERS:
//
//   * slot (iy, ix) owns the window centres [sy0 + iy*fh, +fh) x [sx0 + ix*fw, +fw): exactly
//     one lattice point of every colour class, so the windows of one colour -- real spaxels and
//     the virtual positions that only apply pending updates -- still tile the part's domain.
//     The slot grid is aligned with the colour order: along a row of colour classes
//     (cy fixed, cx = 0 .. fw-1) a slot's window moves right by ONE column per colour;
//   * a thread group (HL threads, one z-pair each) holds one absolute COLUMN of the window,
//     fh rows of residual and 1/variance, in registers across colours.  When the window
//     slides, ten of its eleven columns stay where they are: only the entering column is
//     loaded and only the leaving one stored -- 1/11 of the traffic of a colour launch.  The
//     registers stay current because every update of a cell is applied to them: the
//     predecessor colour's G rows (the slot's own among them) are applied in registers,
//     exactly as k_mh_ws applies a pending layer; memory is brought up to date for the cells
//     another workgroup reads next (the leaving column), and for all of them where the next
//     colour starts a new row of colours (everybody reloads) or the launch ends;
//   * a window of colour k depends on the <= 4 windows of colour k-1 that intersect it, through
//     two monotonic epoch flags per slot: flag1 = "residual stores of colour E complete"
//     and flag2 = "G row of colour E published".  The critical path of a colour is
//     decision -> G row -> flag -> G rows -> apply + accumulate from registers -> decision;
//   * everything that does not depend on the window is off that path: the proposals of a
//     whole sweep (Philox, tan, log: they depend on nothing the sweep changes) are computed
//     at its start, one colour per thread; position table, unit lines and their LSF
//     convolution of colour k+1 are built by the streaming wavefronts while the extra
//     wavefront takes colour k's decision;
//   * hand-off (cdna_hip_programming.md Guideline 16, the all-sc1 form): every residual and
//     G-row byte that crosses workgroups is stored write-through (sc1) and loaded sc1; every
//     storing wavefront drains (vmcnt(0)) and counts itself in LDS, the last one raises the
//     flag (an sc1 store); every wavefront that loads handed-off bytes polls the flags itself.
//
// EXPERIMENTS build only (make EXPERIMENTS=1, option mh_chain = 1).  Measured on MI355X
// (profiles/r03_chain_phases.txt): 12-14 us per colour class against 12.3 us for one colour
// launch at 64^3, an 8x1 rank of 300x300x128 3.6 ms per sweep against 3.5 -- it does NOT beat
// the launches it replaces.  What it removes (the kernel boundary, the window's trip through
// one CU's memory port) it pays back: one CU does a window's whole arithmetic (11 wavefronts
// on 4 SIMDs: 3.2 us), the channel sums and the decision (1.7 + 2.3-3.5 us) stay serial per
// window, and the neighbour's G row reaches the entering column 1.5-5 us after the decision.
//
// Window sums are grouped by window COLUMN (the thread group that holds it accumulates its
// fh rows top to bottom; the columns are then added left to right): another grouping than
// the colour launches' (position p -> group p mod G), so the two agree to rounding, not bit
// for bit -- like every other regrouping of these sums, both match the oracle to 1e-9
// (tests/test_gpu_chain.py).  The chain kernel itself is deterministic and independent of
// how sweeps are cut into launches and of tiling (same parts, same bits).  Every spin has a
// wall-clock bound that raises *F.err and lets the grid drain; the host launches the kernel
// only when all slots are resident at once (one workgroup per CU).
struct MHChain {
    const int2 *cols;   // [K] LOCAL residues (ly, lx) of the part's active colours, in sweep order
    unsigned *flag1;    // [slots] epoch up to which the slot's residual stores are complete
    unsigned *flag2;    // [slots] epoch up to which the slot's G rows are published
    unsigned *err;      // sticky: a wait timed out
    double *G;          // [2][K][slots][Dp] G rows by (sweep parity, colour ordinal, slot)
    double *Gout;       // the LAST colour's rows in the library's regular indexing
                        // ((y/fh)*slots_x + x/fw): the pending layer the launch leaves
    int K, n_sy, n_sx;  // active colours; slot grid
    int sy0, sx0;       // window centres of slot (0,0) start here (sx0 aligned with colour cx = 0)
    int py0, py1, px0, px1;  // the part's rectangle: its unmasked spaxels are the real ones
    unsigned base;      // every flag holds `base` when the launch starts
    uint32_t sweep0;    // Philox sweep number of the first sweep
    int n_sweeps;
    int NS;             // threads that hold a column: fw thread groups of HL threads
    double *lines;      // [slots][K][2][Dp] scratch: LSF-convolved unit lines of a sweep's colours
    int dbg;            // EXPERIMENTS builds: timing-only switches (wrong results), else 0
};

__device__ __forceinline__ int covering_lattice(int q, int c, int per, int hw) {
    int m = (q - c) % per;
    if (m < 0) m += per;
    int s = q - m;
    if (q - s > hw) s += per;
    return s;
}

// spin until *addr has reached epoch `want` (monotonic, wrap-safe); every active lane polls
// its own word.  false: timed out (2 s) or another workgroup did -- *err is set.
__device__ __forceinline__ bool chain_wait(const unsigned *addr, unsigned want, unsigned *err) {
    const unsigned long long t0 = wall_clock64();
    for (unsigned spins = 1;; ++spins) {
        const unsigned v = __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((int)(v - want) >= 0) return true;
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 63) == 0) {
            if (wall_clock64() - t0 > 200000000ULL) {  // 2 s at 100 MHz: give up, let the grid drain
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
            if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
        }
    }
}

constexpr int MH_PROP_DOUBLES = (sizeof(MHProposal) + 7) / 8;
constexpr int MH_CHAIN_GEO = 12;    // ints of a colour's geometry at a slot (k_mh_chain)
constexpr int MH_CHAIN_CHUNK = 8;   // colours whose unit lines are staged in LDS at a time

// LDS of k_mh_chain, in doubles: taps | 4 staged G rows | column partial sums | unit lines of
// a chunk of colours | wave sums + verdict | proposals of a sweep | parameters after each
// colour's update | geometry of each colour | control words
__host__ __device__ inline size_t mh_chain_lds_doubles(int fw, int Dp, int N, int npos, int K,
                                                       int nwaves) {
    return (size_t)npos + 1 + (size_t)Dp + (size_t)fw * 3 * Dp + (size_t)MH_CHAIN_CHUNK * 2 * N +
           8 * (size_t)nwaves + 8 + (size_t)K * (MH_PROP_DOUBLES + 3) +
           ((size_t)K * MH_CHAIN_GEO + 1) / 2 + 4;
}

// lattice point of residue r in [lo, lo + per)
__device__ __forceinline__ int chain_lattice_point(int lo, int r, int per) {
    int m = (r - lo) % per;
    if (m < 0) m += per;
    return lo + m;
}

#ifdef D3D_EXPERIMENTS
#define D3D_CHAIN_STAMP(j)                                                             \
    do {                                                                               \
        if (P.stamp && si == F.n_sweeps - 1 && tid == 0)                               \
            P.stamp[((long)blockIdx.x * F.K + k) * 8 + (j)] = wall_clock64();          \
    } while (0)
#else
#define D3D_CHAIN_STAMP(j)
#endif

// FH = fh, the window's rows: a thread holds FH z-pairs of residual and of 1/variance.
// NTMAX: upper bound of the workgroup size (fw*HL rounded up to wavefronts + 64), which sets
// the register budget.
template <int FH, bool UV, int NTMAX>
__global__ __launch_bounds__(NTMAX) void k_mh_chain(MHArgs P, MHChain F) {
    extern __shared__ double smem[];
    const int tid0 = threadIdx.x;
    int tid = tid0, lane = tid & 63, wave = tid >> 6;
    const int HL = P.HL, Dp = P.Dp, N = P.N, npos = P.npos;
    const int fh = FH, fw = P.fw, fhh = (FH - 1) / 2, fhw = (fw - 1) / 2;
    const int NS = F.NS;                   // threads that hold a column
    const int NW = (NS + 63) / 64;         // streaming wavefronts 0 .. NW-1; wavefront NW decides
    const int NT = NW * 64 + 64;
    MHShared S;
    S.fsf = smem;                                     // [npos + 1]: the last entry is a zero tap
    S.pos = nullptr;
    S.gp = nullptr;
    S.G = S.fsf + npos + 1;                           // [Dp] this slot's own latest G row
    S.red = S.G + Dp;
    S.gO = S.red + (size_t)fw * 3 * Dp;               // [CHUNK][2][N] unit lines of a chunk of colours
    S.gN = nullptr;
    S.sum = S.gO + (size_t)MH_CHAIN_CHUNK * 2 * N;
    double *xb = S.sum + 8 * (size_t)NW + 8;
    MHProposal *sprop = reinterpret_cast<MHProposal *>(xb);   // [K]
    double *snew = xb + (size_t)F.K * MH_PROP_DOUBLES;        // [K][3]
    int *sgeo = reinterpret_cast<int *>(snew + 3 * (size_t)F.K);  // [K][MH_CHAIN_GEO]
    unsigned *sctl = reinterpret_cast<unsigned *>(snew + 3 * (size_t)F.K +
                                                  ((size_t)F.K * MH_CHAIN_GEO + 1) / 2);
    bool streamer = tid < NW * 64;         // (threads NS .. NW*64-1 hold no column)
    int gcol = tid / HL, zl = tid - gcol * HL;  // column group, z-pair
    bool has_column = tid < NS;
    const int slots = F.n_sy * F.n_sx;
    const int slot = blockIdx.x;
    const int iy = slot / F.n_sx, ix = slot - iy * F.n_sx;
    const int ylo = F.sy0 + iy * fh, xlo = F.sx0 + ix * fw;
    const int nstore = (Dp + 63) / 64;  // wavefronts that store the G row
    // LSF-convolved unit lines E_old, E_new of every colour of the sweep, this slot's
    double *lines = F.lines + (size_t)slot * F.K * 2 * Dp;

    for (int p = tid; p <= npos; p += NT) S.fsf[p] = (p < npos) ? P.fsf[p] : 0.0;
    if (tid < 4) sctl[tid] = 0u;
    if (tid < Dp) S.G[tid] = 0.0;

    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    union { double2 d; v4u i; } cv;
    // raw buffer over SLOT_ERR (the launcher checks that it is < 2 GiB); aux 16 = sc1
    const __amdgpu_buffer_rsrc_t err_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        P.err, 0, (int)((long)P.H * P.W * Dp * 8), 0x00020000);

    // Geometry of a colour at this slot (uniform over the workgroup; computed once per
    // sweep by one thread per colour, kept in LDS, read into scalar registers per colour).
    struct Geo {
        int y, x;
        int flags;  // 1 present (the window meets the domain) | 2 real | 4 a predecessor colour exists
        int pk, pring;               // the predecessor colour's ordinal and ring half
        int py0, py1, px0, px1;      // its lattice points at the window's corners (clipped to the domain)
    };
    auto geometry = [&](int si, int k) {
        Geo c;
        const int2 col = F.cols[k];
        c.y = chain_lattice_point(ylo, col.x, fh);
        c.x = chain_lattice_point(xlo, col.y, fw);
        // (the window reaches the domain's upper/left edge by construction of the slot grid,
        // except in the extra slot column the alignment of the grid may add on the left)
        const bool present = (c.y - fhh < P.dy1) && (c.x - fhw < P.dx1) && (c.x + fhw >= P.dx0);
        bool real = false;
        if (present && c.y >= F.py0 && c.y < F.py1 && c.x >= F.px0 && c.x < F.px1)
            real = P.mask[c.y * P.W + c.x] != 0;
        c.flags = (present ? 1 : 0) | (real ? 2 : 0) | ((si | k) ? 4 : 0);
        c.pk = k ? k - 1 : F.K - 1;
        c.pring = k ? (si & 1) : ((si + 1) & 1);
        const int2 pc = F.cols[c.pk];
        c.py0 = covering_lattice(max(c.y - fhh, P.dy0), pc.x, fh, fhh);
        c.py1 = covering_lattice(min(c.y + fhh, P.dy1 - 1), pc.x, fh, fhh);
        c.px0 = covering_lattice(max(c.x - fhw, P.dx0), pc.y, fw, fhw);
        c.px1 = covering_lattice(min(c.x + fhw, P.dx1 - 1), pc.y, fw, fhw);
        return c;
    };
    auto load_geo = [&](int k) {
        const int *gq = sgeo + k * MH_CHAIN_GEO;
        Geo c;
        c.y = __builtin_amdgcn_readfirstlane(gq[0]);
        c.x = __builtin_amdgcn_readfirstlane(gq[1]);
        c.flags = __builtin_amdgcn_readfirstlane(gq[2]);
        c.pk = __builtin_amdgcn_readfirstlane(gq[3]);
        c.pring = __builtin_amdgcn_readfirstlane(gq[4]);
        c.py0 = __builtin_amdgcn_readfirstlane(gq[5]);
        c.py1 = __builtin_amdgcn_readfirstlane(gq[6]);
        c.px0 = __builtin_amdgcn_readfirstlane(gq[7]);
        c.px1 = __builtin_amdgcn_readfirstlane(gq[8]);
        return c;
    };
    // slot of a lattice point of the slot grid
    auto slot_of = [&](int sy, int sx) { return ((sy - F.sy0) / fh) * F.n_sx + (sx - F.sx0) / fw; };

    // the window: column X of this thread group, FH rows, one z-pair per thread
    double2 e[FH], v[FH];
#pragma unroll
    for (int r = 0; r < FH; ++r) {
        e[r] = make_double2(0.0, 0.0);
        // the padding channel of an odd depth carries 1/var = 0
        v[r] = make_double2(P.ivar_uniform, (2 * zl + 1 < P.D) ? P.ivar_uniform : 0.0);
    }
    int held_xl = 0, held_y = 0;
    bool held = false;  // the registers hold the window rows of held_y, columns [held_xl, held_xl + fw)
    int prev_y = -(1 << 30), prev_x = -(1 << 30);  // this slot's lattice point of the previous colour

    bool ok = true;
    for (int si = 0; si < F.n_sweeps; ++si) {
        const uint32_t sweep = F.sweep0 + (uint32_t)si;
        // ---- sweep start.  The 3 x 3 neighbourhood has completed the previous sweep: none of
        // its workgroups still reads a G row of the ring half this sweep overwrites.
        if (si > 0 && wave == 0 && lane < 9) {
            const int ny = iy + lane / 3 - 1, nx = ix + lane % 3 - 1;
            if (ny >= 0 && ny < F.n_sy && nx >= 0 && nx < F.n_sx)
                ok = chain_wait(F.flag2 + ny * F.n_sx + nx, F.base + (unsigned)(si * F.K), F.err);
        }
        // Geometry and proposal of every colour, one colour per thread: they depend on the
        // spaxel's own parameters and its Philox stream only (lib/run.py:369-388) -- on
        // nothing this sweep changes before the colour's turn.
        for (int k2 = tid; k2 < F.K; k2 += NT) {
            const Geo c = geometry(si, k2);
            int *gq = sgeo + k2 * MH_CHAIN_GEO;
            gq[0] = c.y;
            gq[1] = c.x;
            gq[2] = c.flags;
            gq[3] = c.pk;
            gq[4] = c.pring;
            gq[5] = c.py0;
            gq[6] = c.py1;
            gq[7] = c.px0;
            gq[8] = c.px1;
            if (c.flags & 2) {
                const long sp = (long)c.y * P.W + c.x;
                const uint32_t gsp = (uint32_t)((c.y + P.gy0) * P.Wg + (c.x + P.gx0));
                const double a = si ? snew[3 * k2 + 0] : P.params[sp * 3 + 0];
                const double cc = si ? snew[3 * k2 + 1] : P.params[sp * 3 + 1];
                const double w = si ? snew[3 * k2 + 2] : P.params[sp * 3 + 2];
                sprop[k2] = mh_propose_from(P, a, cc, w, gsp, sweep);
            }
        }
        if (__syncthreads_or(!ok)) return;  // a timed-out wait: *F.err is set, the host reports it
        // The LSF-convolved unit lines of every colour (lib/line_models.py:109,
        // lib/convolution.py:89-120), a chunk of colours at a time through LDS, into this
        // slot's scratch rows; the colour loop reads them back with sc1 loads.
        for (int k0 = 0; k0 < F.K; k0 += MH_CHAIN_CHUNK) {
            const int nk = min(MH_CHAIN_CHUNK, F.K - k0);
            for (int i = tid; i < nk * N; i += NT) {
                const int kk = i / N, j = i - kk * N;
                double go = 0.0, gn = 0.0;
                if ((sgeo[(k0 + kk) * MH_CHAIN_GEO + 2] & 2) && j < P.D) {
                    const MHProposal &q = sprop[k0 + kk];
                    go = unit_gaussian((double)j, q.c_old, q.w_old);
                    gn = unit_gaussian((double)j, q.pn[1], q.pn[2]);
                }
                S.gO[(size_t)kk * 2 * N + j] = go;
                S.gO[(size_t)kk * 2 * N + N + j] = gn;
            }
            __syncthreads();
            for (int i = tid; i < nk * Dp; i += NT) {
                const int kk = i / Dp, ch = i - kk * Dp;
                if (sgeo[(k0 + kk) * MH_CHAIN_GEO + 2] & 2) {
                    double EO, EN;
                    mh_lsf(P, S.gO + (size_t)kk * 2 * N, S.gO + (size_t)kk * 2 * N + N, ch, &EO, &EN);
                    double *dst = lines + (size_t)(k0 + kk) * 2 * Dp + ch;
                    dst[0] = EO;
                    dst[Dp] = EN;
                }
            }
            __syncthreads();
        }
        // (the lines were stored by other wavefronts of this workgroup: drained here, read
        // past the L1 below)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        Geo cur = load_geo(0);
        double EO = 0.0, EN = 0.0;  // this thread's channel of the current colour's lines
        if ((cur.flags & 2) && tid < Dp) {
            EO = __longlong_as_double((long long)__hip_atomic_load(
                reinterpret_cast<const unsigned long long *>(lines + tid), __ATOMIC_RELAXED,
                __HIP_MEMORY_SCOPE_AGENT));
            EN = __longlong_as_double((long long)__hip_atomic_load(
                reinterpret_cast<const unsigned long long *>(lines + Dp + tid), __ATOMIC_RELAXED,
                __HIP_MEMORY_SCOPE_AGENT));
        }

        for (int k = 0; k < F.K; ++k) {
            const unsigned E = F.base + (unsigned)(si * F.K + k) + 1u;
            // (the thread index made opaque per colour: what derives from it is recomputed
            // here instead of being hoisted into registers that live for the whole kernel)
            tid = tid0;
            asm volatile("" : "+v"(tid));
            lane = tid & 63;
            wave = tid >> 6;
            streamer = tid < NW * 64;
            gcol = tid / HL;
            zl = tid - gcol * HL;
            has_column = tid < NS;
            const bool present = cur.flags & 1, real = cur.flags & 2, have_pred = cur.flags & 4;
            const bool last = (k + 1 == F.K) && (si + 1 == F.n_sweeps);
            // the next colour of THIS sweep (a sweep's last colour is followed by a reload)
            const bool more = k + 1 < F.K;
            const Geo nxt = more ? load_geo(k + 1) : cur;
            D3D_CHAIN_STAMP(0);
            const int xl = cur.x - fhw, yt = cur.y - fhh;    // the window's left column, top row
            // ---- per ROW of the window, uniform over the workgroup (scalar registers): is it
            // inside the domain; which of the previous colour's two lattice rows (py0 above,
            // py1 below) updated it, and with which tap row
            unsigned in_rows = 0u, upd_rows = 0u;
            const bool sy0_ok = have_pred && cur.py0 >= 0 && cur.py0 < P.H;
            const bool sy1_ok = have_pred && cur.py1 >= 0 && cur.py1 < P.H;
            const int rb = cur.py0 + fhh - yt + 1;           // rows r >= rb belong to py1
#pragma unroll
            for (int r = 0; r < FH; ++r) {
                const int yy = yt + r;
                const bool in = yy >= P.dy0 && yy < P.dy1;
                in_rows |= (in ? 1u : 0u) << r;
                upd_rows |= ((in && (r >= rb ? sy1_ok : sy0_ok)) ? 1u : 0u) << r;
            }
            // ---- this thread group's column: X = xl + ((gcol - xl) mod fw), window column dx
            int dx = (gcol - xl) % fw;
            if (dx < 0) dx += fw;
            const int X = xl + dx;
            const bool col_in = X >= P.dx0 && X < P.dx1;     // inside the part's domain
            const bool slide = held && held_y == cur.y && xl > held_xl && xl - held_xl < fw;
            // with a slide, columns held_xl + fw .. xl + fw - 1 enter: dx >= fw - (xl - held_xl)
            const bool enter = present && has_column && col_in && (!slide || dx >= fw - (xl - held_xl));
            // the previous colour's update of column X comes from lattice column sx: the two
            // spaxels (py0, sx), (py1, sx).  They are this slot's OWN previous spaxel (its G
            // row is in LDS, no hand-off) or another slot's (flags, then sc1 loads).
            const bool right = X - cur.px0 > fhw;
            const int sx = right ? cur.px1 : cur.px0;
            const bool sx_ok = have_pred && sx >= 0 && sx < P.W;
            const int tapx = X - sx + fhw;
            const bool own0 = sx == prev_x && cur.py0 == prev_y;
            const bool own1 = sx == prev_x && cur.py1 == prev_y;
            double2 gz0 = make_double2(0.0, 0.0), gz1 = gz0;
            if (present && has_column && col_in && have_pred) {
                const bool need0 = sx_ok && sy0_ok && (upd_rows & ((1u << min(max(rb, 0), FH)) - 1u));
                const bool need1 = sx_ok && sy1_ok && (upd_rows >> min(max(rb, 0), FH));
                const bool rem0 = (need0 || enter) && !own0, rem1 = (need1 || enter) && !own1;
                // (1) residual stores of the slots that held this column's cells: flag1
                // (one lane per thread group and per wavefront polls: every wavefront that
                // loads handed-off bytes has seen the flags itself)
                const bool poller = zl == 0 || lane == 0;
                if (enter && (rem0 || rem1)) {
                    if (poller && rem0) ok = chain_wait(F.flag1 + slot_of(cur.py0, sx), E - 1u, F.err);
                    if (poller && rem1 && cur.py1 != cur.py0)
                        ok = ok && chain_wait(F.flag1 + slot_of(cur.py1, sx), E - 1u, F.err);
                }
                // (2) their G rows: flag2, then this thread's z-pair of each (sc1)
                if (poller && rem0 && need0) ok = ok && chain_wait(F.flag2 + slot_of(cur.py0, sx), E - 1u, F.err);
                if (poller && rem1 && need1 && cur.py1 != cur.py0)
                    ok = ok && chain_wait(F.flag2 + slot_of(cur.py1, sx), E - 1u, F.err);
                ok = __all(ok);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // keep the loads below the polls
                const double *Gp = F.G + ((size_t)cur.pring * F.K + cur.pk) * slots * Dp;
                auto load_row = [&](int sy) {
                    const unsigned long long *src = reinterpret_cast<const unsigned long long *>(
                        Gp + (size_t)slot_of(sy, sx) * Dp + 2 * zl);
                    double2 g2;
                    g2.x = __longlong_as_double((long long)__hip_atomic_load(src, __ATOMIC_RELAXED,
                                                                             __HIP_MEMORY_SCOPE_AGENT));
                    g2.y = __longlong_as_double((long long)__hip_atomic_load(src + 1, __ATOMIC_RELAXED,
                                                                             __HIP_MEMORY_SCOPE_AGENT));
                    return g2;
                };
                if (need0) gz0 = own0 ? *reinterpret_cast<const double2 *>(S.G + 2 * zl) : load_row(cur.py0);
                if (need1) gz1 = own1 ? *reinterpret_cast<const double2 *>(S.G + 2 * zl) : load_row(cur.py1);
            }
            // the columns that enter the window (rows outside the domain: zeros)
            if (enter) {
#pragma unroll
                for (int r = 0; r < FH; ++r) {
                    if ((in_rows >> r) & 1u) {
                        const long idx = ((long)(yt + r) * P.W + X) * Dp + 2 * zl;
                        cv.i = __builtin_amdgcn_raw_buffer_load_b128(err_rsrc, (int)(idx * 8), 0, 16);
                        e[r] = cv.d;
                        if (!UV) v[r] = *reinterpret_cast<const double2 *>(P.ivar + idx);
                    } else {
                        e[r] = make_double2(0.0, 0.0);
                    }
                }
            }
            held = present;
            held_xl = xl;
            held_y = cur.y;
#ifdef D3D_CHAIN_CLOCK
            if (P.stamp && si == F.n_sweeps - 1 && tid == 0)
                P.stamp[((long)blockIdx.x * F.K + k) * 8 + 1] = __builtin_amdgcn_s_memtime();
#else
            D3D_CHAIN_STAMP(1);
#endif
            D3D_CHAIN_STAMP(2);
            D3D_CHAIN_STAMP(3);
            // ---- (3) apply the pending layer in registers, accumulate; store what another
            // workgroup reads next: the leaving columns -- or everything where the next colour
            // reloads (a new row of colours, the window leaves the domain, a sweep or the
            // launch ends).  S.fsf[npos] = 0: the tap of "no update here" / "row outside".
            if (present && has_column && col_in) {
                const int nxl = nxt.x - fhw;
                const bool nslide = more && (nxt.flags & 1) && nxt.y == cur.y && nxl > xl && nxl - xl < fw;
                const bool store_col = !nslide || dx < nxl - xl;
                double2 sA = make_double2(0.0, 0.0), sB = sA, sC = sA;
#pragma unroll
                for (int r = 0; r < FH; ++r) {
#ifdef D3D_EXPERIMENTS
                    if (F.dbg & 1) continue;   // timing only: nothing per row
#endif
                    const bool in = (in_rows >> r) & 1u;
                    const bool upd = ((upd_rows >> r) & 1u) && sx_ok;
                    const int trow = (r >= rb) ? (yt + r - cur.py1 + fhh) : (yt + r - cur.py0 + fhh);
                    const double fp = S.fsf[upd ? trow * fw + tapx : npos];
                    const double2 gz = (r >= rb) ? gz1 : gz0;
                    e[r].x = fma(fp, gz.x, e[r].x);
                    e[r].y = fma(fp, gz.y, e[r].y);
#ifdef D3D_EXPERIMENTS
                    if (F.dbg & 2) continue;   // timing only: no stores, no sums
#endif
                    if (in && store_col) {
                        cv.d = e[r];
                        __builtin_amdgcn_raw_buffer_store_b128(
                            cv.i, err_rsrc, (int)((((long)(yt + r) * P.W + X) * Dp + 2 * zl) * 8), 0, 16);
                    }
#ifdef D3D_EXPERIMENTS
                    if (F.dbg & 4) continue;   // timing only: no sums
#endif
                    const double f = S.fsf[in ? r * fw + dx : npos];
                    D3D_ACCUM(e[r], v[r], f);
                }
                if (real) {  // partial sums of window column dx
                    double *rr = S.red + (size_t)dx * 3 * Dp + 2 * zl;
                    rr[0] = sA.x;
                    rr[1] = sA.y;
                    rr[Dp] = sB.x;
                    rr[Dp + 1] = sB.y;
                    rr[2 * Dp] = sC.x;
                    rr[2 * Dp + 1] = sC.y;
                }
            } else if (present && has_column && real) {  // a column outside the domain: no terms
                double *rr = S.red + (size_t)dx * 3 * Dp + 2 * zl;
                rr[0] = rr[1] = rr[Dp] = rr[Dp + 1] = rr[2 * Dp] = rr[2 * Dp + 1] = 0.0;
            }
            D3D_CHAIN_STAMP(4);
            if (__syncthreads_or(!ok)) return;  // B1: column partial sums are in S.red (or a wait timed out)
            D3D_CHAIN_STAMP(5);
            // (the wavefronts that hold channels; the decision adds up their sums only)
            if (real && tid < nstore * 64) mh_channel_sums(P, S, sprop[k], tid, fw, EO, EN, 0);
            __syncthreads();  // B2: the wave sums are in S.sum
            D3D_CHAIN_STAMP(6);
            // ---- (4) the extra wavefront decides; meanwhile the streaming wavefronts drain
            // their residual stores (flag1) and fetch the next colour's lines
            double EOn = 0.0, ENn = 0.0;
            if (!streamer) {
#ifdef D3D_EXPERIMENTS
                if (real && !(F.dbg & 8)) {   // (8: timing only, no decision)
#else
                if (real) {
#endif
                    const MHProposal q = sprop[k];
                    const U2 u_gibbs = philox_pair(P.seed, q.gsp, sweep, BLK_GIBBS);
                    mh_decide_wave(P, S, q, cur.y * P.W + cur.x, sweep, nstore, u_gibbs);
                    if (lane == 0) {  // the state the next sweep's proposal starts from
                        const double *verdict = S.sum + 8 * nstore;
                        const bool accept = verdict[0] != 0.0;
                        snew[3 * k + 0] = verdict[1];
                        snew[3 * k + 1] = accept ? q.pn[1] : q.c_old;
                        snew[3 * k + 2] = accept ? q.pn[2] : q.w_old;
                    }
                }
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's sc1 stores
                if (lane == 0) {
                    const unsigned old = __hip_atomic_fetch_add(&sctl[0], 1u, __ATOMIC_RELAXED,
                                                                __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (old == (unsigned)(NW - 1)) {  // the last wavefront to drain signals
                        __hip_atomic_store(&sctl[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_store(F.flag1 + slot, E, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if (more && (nxt.flags & 2) && tid < Dp) {
                    const double *ln = lines + (size_t)(k + 1) * 2 * Dp;
                    EOn = __longlong_as_double((long long)__hip_atomic_load(
                        reinterpret_cast<const unsigned long long *>(ln + tid), __ATOMIC_RELAXED,
                        __HIP_MEMORY_SCOPE_AGENT));
                    ENn = __longlong_as_double((long long)__hip_atomic_load(
                        reinterpret_cast<const unsigned long long *>(ln + Dp + tid), __ATOMIC_RELAXED,
                        __HIP_MEMORY_SCOPE_AGENT));
                }
            }
            __syncthreads();  // B3: the verdict
            D3D_CHAIN_STAMP(7);
            // ---- (5) the G row: into LDS for this slot's own next window, and published
            if (tid < nstore * 64) {
                if (tid < Dp) {
                    const double Gt = real ? mh_update_coeff(P, S, sprop[k], tid, EO, EN, nstore) : 0.0;
                    S.G[tid] = Gt;
                    if (present) {
                        double *dst = F.G + (((size_t)(si & 1) * F.K + k) * slots + slot) * Dp + tid;
                        __hip_atomic_store(reinterpret_cast<unsigned long long *>(dst),
                                           (unsigned long long)__double_as_longlong(Gt), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                        // the pending layer this launch leaves, where flush_pending and the next
                        // launch look for it (in-cube lattice points only, as mh_ws_zero_row)
                        if (last && cur.y >= 0 && cur.y < P.H && cur.x >= 0 && cur.x < P.W)
                            F.Gout[((long)(cur.y / fh) * P.slots_x + cur.x / fw) * Dp + tid] = Gt;
                    }
                }
            }
            __syncthreads();  // B4: this slot's own G row is in LDS
            if (tid < nstore * 64) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) {
                    const unsigned old = __hip_atomic_fetch_add(&sctl[1], 1u, __ATOMIC_RELAXED,
                                                                __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (old == (unsigned)(nstore - 1)) {
                        __hip_atomic_store(&sctl[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_store(F.flag2 + slot, E, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            prev_y = cur.y;
            prev_x = cur.x;
            cur = nxt;
            EO = EOn;
            EN = ENn;
        }
    }
}

#endif  // D3D_EXPERIMENTS (k_mh_chain)

#ifdef D3D_EXPERIMENTS
// ---- one launch per sweep: dataflow over the colour classes ----------------
//
// EXPERIMENTS build only (make EXPERIMENTS=1, option mh_flow = 1).  Measured on MI355X at 300x300x128 / 11x11: 54.9 us
// per colour against 48.4 us for one k_mh_ws launch per colour (uniform
// variance: 49.0 vs 39.1) -- the chain colour k -> k+1 is serial per window, so
// every microsecond of hand-off latency (ticket, flag poll, write-through drain)
// is on the critical path, and only the ~240 workgroup slots a colour leaves
// free can be staged ahead.  Kept because it is bit-identical to the default and
// exercises the in-launch hand-off at full size (tests/test_gpu_full_size.py).
//
// k_mh_ws pays for the kernel boundary after every colour: all workgroups set
// up at the same moment, the slowest window of the colour holds up the next
// launch, and the L2 write-back at the end of the kernel is exposed
// (tools/mh_phases.py: ~13 of 48 us per colour at 300x300x128).  But a window
// of colour k only depends on the <= 4 windows of colour k-1 that intersect it
// (they wrote its residual voxels and hold the G rows it must apply).
// k_mh_flow runs a whole sweep in one launch, one workgroup per window: each
// draws a ticket (items in colour order), waits for the completion flags of
// exactly those predecessors, and publishes its own.  Tickets are handed out in
// dependency order and a workgroup only waits for LOWER tickets, which are
// held by workgroups that are already running -- the lowest unfinished ticket
// can always proceed.  Every spin has a wall-clock timeout that raises
// *F.err and lets the grid drain.
//
// The XCDs' L2s are not coherent with each other and a CU's L1 is never
// refreshed by another CU's stores.  Release/acquire fences per item (L2
// write-back, L1 invalidate: ~7 us each with four workgroups per CU) sit on the
// colour-to-colour critical path and cost more than the kernel boundary they
// replace (measured: 76 vs 48 us per colour).  Instead every byte that is handed
// over inside the launch -- the residual and the G rows -- is stored
// write-through (sc1) and loaded sc1; the storing waves drain (vmcnt(0)) before
// one lane raises the flag.
// G rows cycle through three buffers (colour k writes buffer (pb+k+1) mod 3 and
// reads (pb+k) mod 3); an item of colour k also waits until colour k-2 is
// complete, so that no reader of the buffer it overwrites is still running.
struct MHFlow {
    const int4 *ent;    // [tickets] {y, x, real, ordinal of the colour among the active ones}
    const int4 *col;    // [K] {first ticket of the colour, cy, cx, -}
    const int *lat;     // [K][LY*LX] lattice point -> index in the colour's list, -1 = none
    unsigned *done;     // [tickets] epoch in which the item was finished
    unsigned *cnt;      // [K] finished items of each colour (this launch)
    unsigned *ctl;      // [0] next ticket
    unsigned *err;      // sticky: a dependency wait timed out
    double *gbuf[3];
    int K, LY, LX, pb, items;
    unsigned epoch;
};

__device__ __forceinline__ bool flow_wait(const unsigned *addr, unsigned want, bool at_least,
                                          unsigned *err) {
    const unsigned long long t0 = wall_clock64();
    for (;;) {
        const unsigned v = __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (at_least ? (v >= want) : (v == want)) return true;
        __builtin_amdgcn_s_sleep(8);
        if (wall_clock64() - t0 > 200000000ULL) {  // 2 s at 100 MHz: give up, let the grid drain
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return false;
    }
}

template <int NS, bool UV>
__global__ __launch_bounds__(NS + 64) void k_mh_flow(MHArgs P, MHFlow F, uint32_t sweep) {
    extern __shared__ double smem[];
    constexpr int NT = NS + 64;
    const int tid = threadIdx.x;
    const MHShared S = mh_carve(smem, NS, P.HL, P.Dp, P.N, P.npos);
    int *s_item = reinterpret_cast<int *>(smem + mh_ws_lds_doubles(NS, P.HL, P.Dp, P.N, P.npos));
    const int fhh = (P.fh - 1) / 2, fhw = (P.fw - 1) / 2;
    // The ticket, not blockIdx, orders the items: a workgroup that has drawn one is
    // running, whatever order the dispatcher admits workgroups in.
    if (tid == 0)
        *s_item = (int)__hip_atomic_fetch_add(F.ctl, 1u, __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    // uniform over the workgroup: keep it (and all that follows from it) scalar
    const int item = __builtin_amdgcn_readfirstlane(*s_item);
    if (item >= F.items) return;
    const int4 ent = F.ent[item];  // {y, x, real, colour ordinal}
    const int k = ent.w;
    MHWsItem I;
    I.y = ent.x;
    I.x = ent.y;
    I.real = ent.z;
    int prev_off = 0, prev2_off = 0;
    int prev_cy, prev_cx;
    if (k > 0) {
        const int4 pc = F.col[k - 1];
        prev_cy = pc.y;
        prev_cx = pc.z;
        prev_off = pc.x;
        if (k > 1) prev2_off = F.col[k - 2].x;
    } else {
        prev_cy = P.prev_cy;
        prev_cx = P.prev_cx;
    }
    // one pending layer (the previous colour), written back by every item
    I.n_lay = prev_cy >= 0 ? 1 : 0;
    I.write_back = 1;
    I.rev = P.rev ? (k & 1) : 0;  // P.rev: zig-zag enabled; k: the colour's ordinal
#pragma unroll
    for (int j = 0; j < MH_LAYERS; ++j) {
        I.lay_cy[j] = prev_cy;
        I.lay_cx[j] = prev_cx;
        I.lay_G[j] = F.gbuf[(F.pb + k) % 3];
    }
    I.Gcur = F.gbuf[(F.pb + k + 1) % 3];
    const bool idle = !I.real && I.n_lay == 0;  // nothing pending, nothing to do
    bool ok = true;
    mh_ws_preds<1>(P, I);
    if (!idle) mh_ws_table<1>(P, S, I, NT);
    if (k > 0 && tid < 5) {
        // lanes 0..3: the predecessors (lattice points of colour k-1 whose windows
        // intersect this one inside the cube); lane 4: colour k-2 complete
        if (tid < 4) {
            const int wy = (tid >> 1) ? min(I.y + fhh, P.H - 1) : max(I.y - fhh, 0);
            const int wx = (tid & 1) ? min(I.x + fhw, P.W - 1) : max(I.x - fhw, 0);
            const int sy = covering_lattice(wy, prev_cy, P.fh, fhh);
            const int sx = covering_lattice(wx, prev_cx, P.fw, fhw);
            const int iy = (sy - prev_cy) / P.fh + 1, ix = (sx - prev_cx) / P.fw + 1;
            int li = -1;
            if (iy >= 0 && iy < F.LY && ix >= 0 && ix < F.LX)
                li = F.lat[((long)(k - 1) * F.LY + iy) * F.LX + ix];
            if (li >= 0) ok = flow_wait(F.done + prev_off + li, F.epoch, false, F.err);
        } else if (k > 1) {
            ok = flow_wait(F.cnt + (k - 2), (unsigned)(prev_off - prev2_off), true, F.err);
        }
    }
    // every handed-off byte is stored and loaded sc1 (mh_ws_gp_load<true>,
    // mh_ws_run<.., true>): no acquire; only keep the loads below the polls
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // a timed-out wait skips the item (*F.err is set: the host reports it)
    if (__syncthreads_or(!ok)) return;
    if (!idle) {
        MHGpRegs<1, 4> gv;
        mh_ws_gp_load<1, 4, true>(P, I, NT, gv);
        mh_ws_gp_store<1, 4>(P, S, I, NT, gv);
        __syncthreads();
        mh_ws_run<NS, UV, true, 1, 1>(P, S, I, sweep, item);
    } else {
        mh_ws_zero_row<true>(P, I);
    }
    // every storing wave drains its write-through stores, then one lane raises
    // the flag (cdna_hip_programming.md, Guideline 16 R1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_store(F.done + item, F.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(F.cnt + k, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- two colour classes per launch ------------------------------------------------
//
// With two pending layers the launches alternate: N (finds one layer pending, applies
// it in registers, writes nothing) and W (finds two, applies them, writes the residual
// back).  A W launch depends on the N launch before it ONLY through that colour's G
// rows (1 KiB per window): the residual it reads was last written one launch earlier.
// k_mh_pair therefore runs an N colour and the W colour after it in ONE launch:
//   * items = the windows of colour A (N) followed by those of colour B (W), one
//     workgroup each, in blockIdx order;
//   * an A item publishes its G row with agent-scope (sc1) stores and raises its flag;
//   * a B item waits for the flags of the <= 4 A windows that intersect its own, loads
//     their G rows with sc1 loads, and otherwise is a plain k_mh_ws W item: residual
//     loads and stores are ordinary cached accesses.  No race on the residual: the one
//     A window that covers a cell has read it before it raises the flag the covering
//     B window waits for.
// One kernel boundary less per colour pair; bit-identical to the per-colour launches
// (same windows, same arithmetic; tests/test_gpu_parity.py, test_gpu_full_size.py).
// EXPERIMENTS build only (option mh_pair = 1).  Measured on MI355X at 300x300x128 / 11x11: 43.0 us per colour
// against 42.0 for one launch per colour.  The boundary it saves (an empty pair launch
// costs 3.7 us) is paid back as dependency wait: all windows of colour A are resident at
// once, share the HBM stream equally and therefore finish TOGETHER, so no B window can
// start early -- with the waits switched off (wrong results, timing only) the same launch
// takes 38.8 us per colour.  A first version drew tickets from one atomic counter: 1568
// same-address atomics serialise at ~13 ns each, an empty launch took 20 us.
struct MHPair {
    const int4 *ent;   // [items] {y, x, real, colour ordinal} (the context's flow tables)
    const int *lat;    // [K][LY*LX] lattice point -> index in the colour's list
    unsigned *done;    // [items] epoch in which the item finished
    unsigned *ctl;     // ticket counter (monotonic over launches)
    unsigned *err;     // sticky: a wait timed out
    int first_a, n_a;  // items of colour A: [first_a, first_a + n_a)
    int first_b, n_b;  // items of colour B
    int ka;            // ordinal of colour A among the active colours (lat table row)
    int a_cy, a_cx;    // colour class of A
    int LY, LX;
    unsigned ticket_base;  // value of *ctl when this launch starts
    unsigned epoch;
    double *G_a, *G_b;     // G rows written by A items / B items
};

template <int NS, bool UV, int U, int K>
__global__ __launch_bounds__(NS + 64) void k_mh_pair(MHArgs P, MHPair F, uint32_t sweep) {
    extern __shared__ double smem[];
    constexpr int NT = NS + 64;
    constexpr int M = 2;
    const int tid = threadIdx.x;
    const MHShared S = mh_carve(smem, NS, P.HL, P.Dp, P.N, P.npos, M);
    const int fhh = (P.fh - 1) / 2, fhw = (P.fw - 1) / 2;
    // Items in blockIdx order, NOT tickets from an atomic counter: 1568 same-address
    // atomics per launch serialise at ~13 ns each -- an empty launch took 20 us.  The
    // dispatcher hands out the workgroups of a 1-D grid in index order per XCD, and every
    // A item precedes every B item, so an A item is never kept out of a slot by a B item
    // that waits for it; should that ever fail, flow_wait's time-out raises *F.err and
    // the host reports it instead of hanging.
    const int t = (int)blockIdx.x;
    if (t >= F.n_a + F.n_b) return;
#ifdef D3D_EXPERIMENTS
    mh_stagger(P.prio);
#endif
    const bool is_b = t >= F.n_a;
    const int item = is_b ? F.first_b + (t - F.n_a) : F.first_a + t;
    const int4 ent = F.ent[item];
    MHWsItem I;
    I.y = ent.x;
    I.x = ent.y;
    I.real = ent.z;
    // pending layers: those of the launch arguments, and for a B item colour A on top
    mh_ws_layers_from_args(P, I);
    I.rev = P.rev ? ((F.ka + (is_b ? 1 : 0)) & 1) : 0;  // P.rev: zig-zag enabled
    bool ok = true;
    if (is_b) {
        // (the host fuses a pair only when exactly ONE layer is pending: static index 1)
        I.lay_cy[1] = F.a_cy;
        I.lay_cx[1] = F.a_cx;
        I.lay_G[1] = F.G_a;
        I.n_lay = 2;
        I.write_back = 1;
        I.Gcur = F.G_b;
        if (tid < 4) {
            // the A windows that intersect this one inside the domain
            const int wy = (tid >> 1) ? min(I.y + fhh, P.dy1 - 1) : max(I.y - fhh, P.dy0);
            const int wx = (tid & 1) ? min(I.x + fhw, P.dx1 - 1) : max(I.x - fhw, P.dx0);
            const int sy = covering_lattice(wy, F.a_cy, P.fh, fhh);
            const int sx = covering_lattice(wx, F.a_cx, P.fw, fhw);
            const int iy = (sy - F.a_cy) / P.fh + 1, ix = (sx - F.a_cx) / P.fw + 1;
            int li = -1;
            if (iy >= 0 && iy < F.LY && ix >= 0 && ix < F.LX)
                li = F.lat[((long)F.ka * F.LY + iy) * F.LX + ix];
            if (li >= 0) ok = flow_wait(F.done + F.first_a + li, F.epoch, false, F.err);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // keep the loads below the polls
    } else {
        I.write_back = 0;
        I.Gcur = F.G_a;
    }
    if (__syncthreads_or(!ok)) return;  // a timed-out wait (*F.err is set: the host reports it)
    const bool idle = !I.real && (I.n_lay == 0 || !I.write_back);
    if (idle) {
        if (!is_b) mh_ws_zero_row<true>(P, I); else mh_ws_zero_row<false>(P, I);
    } else {
        mh_ws_preds<M>(P, I);
        MHGpRegs<M, K> gv;
        if (is_b)
            mh_ws_gp_load<M, K, true>(P, I, NT, gv);   // colour A's rows come from this launch
        else
            mh_ws_gp_load<M, K, false>(P, I, NT, gv);
        mh_ws_table<M>(P, S, I, NT);
        mh_ws_gp_store<M, K>(P, S, I, NT, gv);
        __syncthreads();
        if (is_b)
            mh_ws_run<NS, UV, false, U, M, false>(P, S, I, sweep, item);
        else
            mh_ws_run<NS, UV, false, U, M, true>(P, S, I, sweep, item);
    }
    if (!is_b) {
        // every storing wave drains its sc1 stores, then one lane raises the flag
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0)
            __hip_atomic_store(F.done + item, F.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

#endif  // D3D_EXPERIMENTS (k_mh_flow, k_mh_pair)

// Replay of updates made by ANOTHER tile (multi-GPU spatial tiling): one
// workgroup per record {ly, lx, a_old, c_old, w_old, a_new, c_new, w_new} with
// (ly, lx) in this tile's local coordinates (possibly outside it).  Recomputes
// the update coefficient exactly as the owner did and applies err += f*G on the
// part of the window that lies inside the local region; spaxels inside the
// region also get their parameters.  Thread t <-> channel t for the lines, then
// z-pairs for the window.
template <int NT>
__global__ __launch_bounds__(NT) void k_apply_updates(MHArgs P, const double *__restrict__ rec,
                                                      int nrec) {
    extern __shared__ double smem[];
    double *gO = smem;          // [N]
    double *gN = gO + P.N;      // [N]
    double *sG = gN + P.N;      // [Dp]
    const int tid = threadIdx.x;
    const double *r8 = rec + (long)blockIdx.x * 8;
    const int y = (int)r8[0], x = (int)r8[1];
    const double a_old = r8[2], c_old = r8[3], w_old = r8[4];
    const double a_new = r8[5], c_new = r8[6], w_new = r8[7];
    for (int j = tid; j < P.N; j += NT) {
        gO[j] = (j < P.D) ? unit_gaussian((double)j, c_old, w_old) : 0.0;
        gN[j] = (j < P.D) ? unit_gaussian((double)j, c_new, w_new) : 0.0;
    }
    __syncthreads();
    for (int ch = tid; ch < P.Dp; ch += NT) {
        double EO, EN;
        mh_lsf(P, gO, gN, ch, &EO, &EN);
        sG[ch] = (ch < P.D) ? residual_coeff(a_old, EO, a_new, EN) : 0.0;
    }
    if (tid == 0 && y >= 0 && y < P.H && x >= 0 && x < P.W) {
        const long sp = (long)y * P.W + x;
        P.params[sp * 3 + 0] = a_new;
        P.params[sp * 3 + 1] = c_new;
        P.params[sp * 3 + 2] = w_new;
    }
    __syncthreads();
    const int HL = P.HL, G = NT / HL;
    const int g = tid / HL, zl = tid - g * HL;
    if (g >= G) return;
    const int fhh = (P.fh - 1) / 2, fhw = (P.fw - 1) / 2;
    const double2 Gz = *reinterpret_cast<const double2 *>(sG + 2 * zl);
    for (int p = g; p < P.npos; p += G) {
        const int dy = p / P.fw, dx = p - dy * P.fw;
        const int yy = y + dy - fhh, xx = x + dx - fhw;
        if (yy < 0 || yy >= P.H || xx < 0 || xx >= P.W) continue;
        const long idx = ((long)yy * P.W + xx) * P.Dp + 2 * zl;
        const double f = P.fsf[p];
        double2 e = *reinterpret_cast<const double2 *>(P.err + idx);
        e.x = fma(f, Gz.x, e.x);
        e.y = fma(f, Gz.y, e.y);
        *reinterpret_cast<double2 *>(P.err + idx) = e;
    }
}

// Records of the last update of the listed spaxels, for a neighbour tile:
// {global y, global x, a_old, c_old, w_old, a_new, c_new, w_new}.
static __global__ void k_gather_updates(MHArgs P, const int *__restrict__ idx, int n,
                                 double *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int sp = idx[i];
    const int y = sp / P.W, x = sp - y * P.W;
    double *o = out + (long)i * 8;
    o[0] = (double)(y + P.gy0);
    o[1] = (double)(x + P.gx0);
    for (int k = 0; k < 3; ++k) {
        o[2 + k] = P.prev[(long)sp * 3 + k];
        o[5 + k] = P.params[(long)sp * 3 + k];
    }
}

// n draws of TN(lo, hi; mu, sigma) with the sampler of the Gibbs step (d3d_rng.h;
// distribution of lib/rtnorm.py:21-92).  Draw i uses the Philox stream of
// (spaxel i, sweep 0).  wave_mode = 0: one thread per draw (scalar form);
// wave_mode = 1: one wavefront per draw, every lane the same arguments -- the
// form mh_finish runs (the two CDFs side by side in the wavefront's halves).
static __global__ __launch_bounds__(256) void k_rtnorm(long n, double lo, double hi, double mu,
                                                 double sigma, uint64_t seed, int wave_mode,
                                                 double *__restrict__ out) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long i = wave_mode ? t >> 6 : t;
    if (i >= n) return;
    uint32_t blk = BLK_GIBBS;
    const U2 u0 = philox_pair(seed, (uint32_t)i, 0u, BLK_GIBBS);
    double r;
    if (wave_mode)
        r = truncated_normal<true>(lo, hi, mu, sigma, u0, seed, (uint32_t)i, 0u, &blk);
    else
        r = truncated_normal<false>(lo, hi, mu, sigma, u0, seed, (uint32_t)i, 0u, &blk);
    if (!wave_mode || (t & 63) == 0) out[i] = r;
}

// Apply the pending layers (oldest first) to the whole residual, before anything
// other than the next colour launch looks at it.
template <int NT>
__global__ __launch_bounds__(NT) void k_flush_pending(MHArgs P) {
    const int dw = P.dx1 - P.dx0;
    int zl;
    long cell;  // cell of the domain, row-major
    if (P.HL > NT) {  // a spectrum longer than the workgroup: blockIdx.y counts its NT-thread pieces
        zl = blockIdx.y * NT + threadIdx.x;
        cell = blockIdx.x;
        if (zl >= P.HL) return;
    } else {
        const int S = NT / P.HL;
        const int s = threadIdx.x / P.HL;
        zl = threadIdx.x - s * P.HL;
        cell = (long)blockIdx.x * S + s;
        if (s >= S) return;
    }
    if (cell >= (long)(P.dy1 - P.dy0) * dw) return;
    const int yy = P.dy0 + (int)(cell / dw), xx = P.dx0 + (int)(cell - (cell / dw) * dw);
    const long vox = (long)yy * P.W + xx;
    const int fhh = (P.fh - 1) / 2, fhw = (P.fw - 1) / 2;
    double2 e = *reinterpret_cast<const double2 *>(P.err + vox * P.Dp + 2 * zl);
    bool touched = false;
#pragma unroll
    for (int j = 0; j < MH_LAYERS; ++j) {
        if (j >= P.n_lay) continue;
        const int sy = covering_coord(yy, P.lay_cy[j], P.fh, fhh, P.H);
        const int sx = covering_coord(xx, P.lay_cx[j], P.fw, fhw, P.W);
        // (k_mh_ws: an unmasked lattice point outside the part being updated left a zero
        // G row, mh_ws_zero_row; k_mh_defer leaves none for masked ones: check the mask)
        if (sy < 0 || sx < 0 || !P.mask[sy * P.W + sx]) continue;
        const double fp = P.fsf[(yy - sy + fhh) * P.fw + (xx - sx + fhw)];
        const double2 gz = *reinterpret_cast<const double2 *>(
            P.lay_G[j] + ((long)(sy / P.fh) * P.slots_x + sx / P.fw) * P.Dp + 2 * zl);
        e.x = fma(fp, gz.x, e.x);
        e.y = fma(fp, gz.y, e.y);
        touched = true;
    }
    if (touched) *reinterpret_cast<double2 *>(P.err + vox * P.Dp + 2 * zl) = e;
}

// ------------------------------------------------------------------------- //
// Deep cubes (more than 1024 channels: a full MUSE cube has ~3700)             //
// ------------------------------------------------------------------------- //
// The kernels above map a spectrum onto the threads of ONE workgroup, a z-pair per
// thread: 1024 channels at most.  The reference takes any depth (lib/convolution.py:
// 137-141 pads to the next power of two; lib/run.py:146-149).  Deeper cubes run the
// forms below: the same arithmetic with a thread looping over its channels
// (z-blocked), plain and unoptimised -- correct first (tests/test_gpu_edges.py up to
// 3700 channels); up to MH_DEEP_MAX channels (the zero-extended unit lines of length
// N = 2^k >= D must fit the LDS twice).
constexpr int MH_DEEP_ZB = 4;               // z-pairs per thread of k_mh_deep (1024 threads)
constexpr int MH_DEEP_MAX = 2 * 1024 * MH_DEEP_ZB;

// k_lines for any depth: one workgroup per spaxel.
static __global__ __launch_bounds__(1024) void k_lines_deep(SpectralArgs A,
                                                            const double *__restrict__ params,
                                                            const uint8_t *__restrict__ mask,
                                                            double *__restrict__ out, int convolved) {
    extern __shared__ double smem[];
    const long sp = blockIdx.x;
    const bool live = mask[sp] != 0;
    const double a = params[sp * 3 + 0], c = params[sp * 3 + 1], w = params[sp * 3 + 2];
    const bool use_lsf = convolved && A.ntaps > 0;
    if (!use_lsf) {
        for (int z = threadIdx.x; z < A.Dp; z += 1024)
            out[sp * A.Dp + z] = (live && z < A.D) ? a * unit_gaussian((double)z, c, w) : 0.0;
        return;
    }
    for (int j = threadIdx.x; j < A.N; j += 1024)
        smem[j] = (live && j < A.D) ? a * unit_gaussian((double)j, c, w) : 0.0;
    __syncthreads();
    for (int z = threadIdx.x; z < A.Dp; z += 1024)
        out[sp * A.Dp + z] = (live && z < A.D) ? lsf_apply(smem, z, A) : 0.0;
}

// k_spectral for any depth.
static __global__ __launch_bounds__(1024) void k_spectral_deep(SpectralArgs A,
                                                               const double *__restrict__ in,
                                                               double *__restrict__ out) {
    extern __shared__ double smem[];
    const long sp = blockIdx.x;
    for (int j = threadIdx.x; j < A.N; j += 1024) smem[j] = (j < A.D) ? in[sp * A.Dp + j] : 0.0;
    __syncthreads();
    for (int z = threadIdx.x; z < A.Dp; z += 1024)
        out[sp * A.Dp + z] = (z < A.D) ? lsf_apply(smem, z, A) : 0.0;
}

// k_spatial_generic for any depth: grid (spaxels, z-chunks of 256 z-pairs).
static __global__ __launch_bounds__(256) void k_spatial_deep(SpatialArgs A,
                                                             const double *__restrict__ in,
                                                             double *__restrict__ out) {
    const long sp = blockIdx.x;
    const int zl = blockIdx.y * 256 + threadIdx.x;
    if (zl >= A.HL) return;
    const int y = (int)(sp / A.W), x = (int)(sp - (long)y * A.W);
    const int fhh = (A.fh - 1) / 2, fhw = (A.fw - 1) / 2;
    double2 acc = make_double2(0.0, 0.0);
    for (int j = 0; j < A.fh; ++j) {
        const int yy = y - j + fhh;
        if (yy < 0 || yy >= A.H) continue;
        for (int i = 0; i < A.fw; ++i) {
            const int xx = x - i + fhw;
            if (xx < 0 || xx >= A.W) continue;
            const double tap = A.fsf[j * A.fw + i];
            const double2 v =
                *reinterpret_cast<const double2 *>(in + ((long)yy * A.W + xx) * A.Dp + 2 * zl);
            acc.x = fma(tap, v.x, acc.x);
            acc.y = fma(tap, v.y, acc.y);
        }
    }
    const long o = sp * A.Dp + 2 * zl;
    if (A.data) {
        const double2 d = *reinterpret_cast<const double2 *>(A.data + o);
        acc.x = d.x - acc.x;
        acc.y = d.y - acc.y;
    }
    *reinterpret_cast<double2 *>(out + o) = acc;
}

__host__ __device__ inline size_t mh_deep_lds_doubles(int N, int npos) {
    return (size_t)npos + 2 * (size_t)N + 8 * 16 + 8;
}

// The terms one channel adds to the seven sums of the decision (mh_channel_sums).
__device__ __forceinline__ void mh_channel_terms(const MHProposal &q, double EO, double EN, double Az,
                                                 double Bz, double Cz, double (&s)[7]) {
    const double a_new = q.pn[0];
    const double Lo = q.a_old * EO;
    const double d = Lo - a_new * EN;
    const double ulB = Az + Lo * Bz;
    s[0] += d * Az;
    s[1] += d * d * Bz;
    s[2] += Cz;
    s[3] += EO * EO * Bz;
    s[4] += EO * ulB;
    s[5] += EN * EN * Bz;
    s[6] += EN * ulB;
}

// One MH-within-Gibbs update per workgroup (lib/run.py:367-519), immediate write-back, any
// depth up to MH_DEEP_MAX: k_mh with every thread looping over its z-pairs tid, tid + 1024,
// ...  The window sums stay in registers (one position group), the zero-extended unit lines
// go to LDS, the decision is mh_decide_wave's.  Probe and external-lines modes as k_mh.
static __global__ __launch_bounds__(1024) void k_mh_deep(MHArgs P, uint32_t sweep) {
    extern __shared__ double smem[];
    constexpr int NT = 1024, ZB = MH_DEEP_ZB;
    const int tid = threadIdx.x;
    const int HL = P.HL, Dp = P.Dp, N = P.N, npos = P.npos;
    const int fhh = (P.fh - 1) / 2, fhw = (P.fw - 1) / 2;
    MHShared S = {};
    S.fsf = smem;
    S.gO = smem + npos;
    S.gN = S.gO + N;
    S.sum = S.gN + N;
    int sp;
    if (P.probe) {
        sp = P.probe_sp;
    } else if (P.ext_lines) {
        sp = P.ext_idx[blockIdx.x];
    } else {
        const int4 ent = P.spx[blockIdx.x];
        sp = ent.x * P.W + ent.y;
    }
    const int y = sp / P.W, x = sp - y * P.W;
    for (int p = tid; p < npos; p += NT) S.fsf[p] = P.fsf[p];
    // the decision's window-independent part first: proposal and zero-extended unit lines
    const MHProposal q = mh_propose(P, sp, sweep);
    for (int j = tid; j < N; j += NT) {
        if (P.ext_lines) {
            const double *L = P.ext_lines + (long)blockIdx.x * 2 * P.D;
            S.gO[j] = (j < P.D) ? L[j] : 0.0;
            S.gN[j] = (j < P.D) ? L[P.D + j] : 0.0;
        } else {
            S.gO[j] = (j < P.D) ? unit_gaussian((double)j, q.c_old, q.w_old) : 0.0;
            S.gN[j] = (j < P.D) ? unit_gaussian((double)j, q.pn[1], q.pn[2]) : 0.0;
        }
    }
    __syncthreads();

    // ---- pass 1: window sums per channel, in registers ---------------------------------
    double2 wA[ZB], wB[ZB], wC[ZB];
#pragma unroll
    for (int j = 0; j < ZB; ++j) wA[j] = wB[j] = wC[j] = make_double2(0.0, 0.0);
    for (int pw = 0; pw < npos; ++pw) {
        const int p = P.rev ? npos - 1 - pw : pw;
        const int dy = p / P.fw, dx = p - dy * P.fw;
        const int yy = y + dy - fhh, xx = x + dx - fhw;
        if (yy < 0 || yy >= P.H || xx < 0 || xx >= P.W) continue;
        const double f = S.fsf[p];
        const long base = ((long)yy * P.W + xx) * Dp;
#pragma unroll
        for (int j = 0; j < ZB; ++j) {
            const int zl = tid + j * NT;
            if (zl < HL) {
                const double2 e = *reinterpret_cast<const double2 *>(P.err + base + 2 * zl);
                const double2 v = *reinterpret_cast<const double2 *>(P.ivar + base + 2 * zl);
                double2 sA = wA[j], sB = wB[j], sC = wC[j];
                D3D_ACCUM(e, v, f);
                wA[j] = sA;
                wB[j] = sB;
                wC[j] = sC;
            }
        }
    }
    // ---- the seven sums over this thread's channels, then over the workgroup ----------------
    double s7[7] = {0, 0, 0, 0, 0, 0, 0};
    double2 EOr[ZB], ENr[ZB];
#pragma unroll
    for (int j = 0; j < ZB; ++j) {
        EOr[j] = ENr[j] = make_double2(0.0, 0.0);
        const int zl = tid + j * NT;
        if (zl < HL) {
            mh_lsf(P, S.gO, S.gN, 2 * zl, &EOr[j].x, &ENr[j].x);
            mh_lsf(P, S.gO, S.gN, 2 * zl + 1, &EOr[j].y, &ENr[j].y);
            if (2 * zl < P.D) mh_channel_terms(q, EOr[j].x, ENr[j].x, wA[j].x, wB[j].x, wC[j].x, s7);
            if (2 * zl + 1 < P.D) mh_channel_terms(q, EOr[j].y, ENr[j].y, wA[j].y, wB[j].y, wC[j].y, s7);
        }
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) s7[k] = wave_sum_dpp63(s7[k]);
    if ((tid & 63) == 63) {
#pragma unroll
        for (int k = 0; k < 7; ++k) S.sum[(tid >> 6) * 8 + k] = s7[k];
    }
    __syncthreads();
    if (tid < 64) {
        const U2 u_gibbs = philox_pair(P.seed, q.gsp, sweep, BLK_GIBBS);
        mh_decide_wave(P, S, q, sp, sweep, NT / 64, u_gibbs);
    }
    __syncthreads();
    if (P.probe) return;

    // ---- pass 2: write the window back --------------------------------------------------
    double2 Gz[ZB];
#pragma unroll
    for (int j = 0; j < ZB; ++j) {
        const int zl = tid + j * NT;
        Gz[j].x = mh_update_coeff(P, S, q, 2 * zl, EOr[j].x, ENr[j].x, NT / 64);
        Gz[j].y = mh_update_coeff(P, S, q, 2 * zl + 1, EOr[j].y, ENr[j].y, NT / 64);
    }
    for (int pw = 0; pw < npos; ++pw) {
        const int p = P.rev ? npos - 1 - pw : pw;
        const int dy = p / P.fw, dx = p - dy * P.fw;
        const int yy = y + dy - fhh, xx = x + dx - fhw;
        if (yy < 0 || yy >= P.H || xx < 0 || xx >= P.W) continue;
        const double f = S.fsf[p];
        const long base = ((long)yy * P.W + xx) * Dp;
#pragma unroll
        for (int j = 0; j < ZB; ++j) {
            const int zl = tid + j * NT;
            if (zl < HL) {
                double2 e = *reinterpret_cast<const double2 *>(P.err + base + 2 * zl);
                e.x = fma(f, Gz[j].x, e.x);
                e.y = fma(f, Gz[j].y, e.y);
                *reinterpret_cast<double2 *>(P.err + base + 2 * zl) = e;
            }
        }
    }
}

// The proposals of one sweep for every unmasked spaxel of the rectangle [y0,y1) x [x0,x1)
// (MHArgs::props), one thread per spaxel.
// Z-blocked sweeps, second half of a colour class: one workgroup per window of the launch.
// Totals the blocks' wave sums (block by block, wave by wave: a fixed order), takes the decision
// (mh_decide_wave: accept, Gibbs draw, state), and forms the window's G row from the lines the
// blocks left in z_E -- the pending layer the next colour's k_mh_ws<..., ZBK> applies.
static __global__ __launch_bounds__(256) void k_mh_zdecide(MHArgs P, uint32_t sweep, int waves_per_block) {
    extern __shared__ double smem[];
    const int tid = threadIdx.x;
    const int4 ent = P.spx[blockIdx.x];
    if (!ent.z) return;  // a virtual position: it only applied pending layers
    const int y = ent.x, x = ent.y, sp = y * P.W + x;
    const int nw = P.z_nb * waves_per_block;
    MHShared S = {};
    S.sum = smem;                                                  // [nw][8] + the verdict
    MHProposal *sq = reinterpret_cast<MHProposal *>(smem + (size_t)8 * nw + 8);
    for (int i = tid; i < nw * 8; i += 256) S.sum[i] = P.z_part[(long)blockIdx.x * nw * 8 + i];
    __syncthreads();
    if (tid < 64) {
        const MHProposal q = mh_proposal_of(P, sp, sweep);
        const U2 u_gibbs = philox_pair(P.seed, q.gsp, sweep, BLK_GIBBS);
        mh_decide_wave(P, S, q, sp, sweep, nw, u_gibbs);
        if (tid == 0) *sq = q;
    }
    __syncthreads();
    const double *verdict = S.sum + 8 * nw;
    const bool accept = verdict[0] != 0.0;
    const double r = verdict[1], a_old = sq->a_old;
    const long slot = (long)(y / P.fh) * P.slots_x + x / P.fw;
    const double *eo = P.z_E + slot * P.Dp, *en = eo + (long)P.z_slots * P.Dp;
    double *g = P.Gcur + slot * P.Dp;
    // (z-pairs, four per thread in flight: the loop is a chain of memory latencies otherwise)
    for (int z0 = 2 * tid; z0 < P.Dp; z0 += 4 * 512) {
        double2 a[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int z = z0 + k * 512;
            a[k] = b[k] = make_double2(0.0, 0.0);
            if (z < P.Dp) {
                a[k] = *reinterpret_cast<const double2 *>(eo + z);
                if (accept) b[k] = *reinterpret_cast<const double2 *>(en + z);
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int z = z0 + k * 512;
            if (z >= P.Dp) continue;
            double2 o;
            o.x = (z < P.D) ? residual_coeff(a_old, a[k].x, r, accept ? b[k].x : a[k].x) : 0.0;
            o.y = (z + 1 < P.D) ? residual_coeff(a_old, a[k].y, r, accept ? b[k].y : a[k].y) : 0.0;
            *reinterpret_cast<double2 *>(g + z) = o;
        }
    }
}

static __global__ __launch_bounds__(256) void k_mh_proposals(MHArgs P, uint32_t sweep, int y0, int y1,
                                                              int x0, int x1, MHProposal *out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int w = x1 - x0;
    if (i >= (y1 - y0) * w) return;
    const int y = y0 + i / w, x = x0 + i % w;
    const long sp = (long)y * P.W + x;
    if (!P.mask[sp]) return;
    out[sp] = mh_propose_from(P, P.params[sp * 3 + 0], P.params[sp * 3 + 1], P.params[sp * 3 + 2],
                              (uint32_t)((y + P.gy0) * P.Wg + (x + P.gx0)), sweep);
}

// Halo exchange of the tiled chain: the cells (all E values per spaxel: E = Dp for
// a cube, 3 for the parameter map) of the rectangle [y0,y1) x [x0,x1) of a (H,W,E)
// array to / from a packed buffer.  A row of the rectangle is one contiguous run.
static __global__ __launch_bounds__(256) void k_rect_copy(double *__restrict__ arr, int W, int E, int y0,
                                                    int x0, int ny, int nx,
                                                    double *__restrict__ packed, int unpack) {
    const long run = (long)nx * E;
    const long total = (long)ny * run;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long r = i / run, o = i - r * run;
        double *cell = arr + ((long)(y0 + r) * W + x0) * E + o;
        if (unpack)
            *cell = packed[i];
        else
            packed[i] = *cell;
    }
}

}  // namespace d3d
