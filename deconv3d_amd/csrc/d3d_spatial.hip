// Launchers of the line-build, LSF and FSF kernels (d3d_kernels.h, d3d_conv.h).  gfx950 only.
#include "d3d_ctx.h"
#include "d3d_conv.h"

namespace d3dh {

using namespace d3d;

d3d::SpectralArgs spectral_args(const d3d_ctx *c) {
    d3d::SpectralArgs A;
    A.D = c->D;
    A.Dp = c->Dp;
    A.HL = c->HL;
    A.N = c->N;
    A.ntaps = c->ntaps;
    A.nspax = c->HW;
    A.shift = c->lsf_shift;
    A.weight = c->lsf_weight;
    return A;
}

// block size for the group-per-spaxel kernels: at least HL threads.
int pick_nt(int HL) {
    if (HL <= 256) return 256;
    if (HL <= 512) return 512;
    return 1024;
}

template <int NT>
int launch_lines_nt(d3d_ctx *c, double *out, int convolved, const double *params) {
    d3d::SpectralArgs A = spectral_args(c);
    const int G = NT / c->HL;
    const unsigned grid = (unsigned)((c->HW + G - 1) / G);
    const size_t lds = (size_t)G * c->N * sizeof(double);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(d3d::k_lines<NT>), dim3(grid), dim3(NT), lds, c->stream, A,
                       params, c->mask, out, convolved);
    HIP_TRY(hipGetLastError());
    return 0;
}

// params: (H,W,3) map on the device (NULL: the chain state c->params)
int launch_lines(d3d_ctx *c, double *out, int convolved, const double *params) {
    if (!params) params = c->params;
    // the LSF applied here (depths whose FSF pass has no LSF epilogue) and its taps within +-8
    // channels: a lane group of one wavefront per spaxel, dense taps, no block barrier --
    // 300x300x256: 114 -> 76 us.  Without the LSF the kernel is bound by the fp64 instructions of
    // its two exp per thread either way (35.2 against 35.9 us at 128 channels, 24.4 against 25.2
    // at 64: profiles/r04_line_kernels.txt) and the tap-list kernel stays.
    const bool use_lsf = convolved && c->ntaps > 0;
    if (!c->deep && ((c->lines_dense == 1 && use_lsf && c->lsf_dense_any) ||
                     (c->lines_dense >= 2 && (!use_lsf || c->lsf_dense_any)))) {
        int hlg = 8;
        while (hlg < 64 && 2 * hlg < c->Dp) hlg *= 2;
        const int steps = (c->Dp + 2 * hlg - 1) / (2 * hlg), S = 64 / hlg;
        const size_t lds = (size_t)4 * S * (steps * 2 * hlg + 2 * d3d::LSF_RL) * sizeof(double);
        if (lds <= 64 * 1024) {
            // rounds per wavefront: up to 8 while the launch keeps >= 4096 wavefronts
            int L = (int)std::min<long>(8, std::max<long>(1, c->HW / ((long)S * 4096)));
            if (c->lines_rounds > 0) L = std::min(c->lines_rounds, 64 / S);  // (a lane per spaxel of the wavefront)
            const unsigned grid = (unsigned)((c->HW + (long)4 * S * L - 1) / ((long)4 * S * L));
            if (c->lines_dense == 2)
                hipLaunchKernelGGL(HIP_KERNEL_NAME(d3d::k_lines_dense<true>), dim3(grid), dim3(256),
                                   use_lsf ? lds : 0, c->stream, spectral_args(c), hlg, steps, L,
                                   (const double *)c->lsf_dense, params, (const uint8_t *)c->mask, out,
                                   convolved);
            else
                hipLaunchKernelGGL(HIP_KERNEL_NAME(d3d::k_lines_dense<false>), dim3(grid), dim3(256),
                                   use_lsf ? lds : 0, c->stream, spectral_args(c), hlg, steps, L,
                                   (const double *)c->lsf_dense, params, (const uint8_t *)c->mask, out,
                                   convolved);
            HIP_TRY(hipGetLastError());
            return 0;
        }
    }
    if (c->deep) {
        d3d::SpectralArgs A = spectral_args(c);
        const size_t lds = (size_t)c->N * sizeof(double);
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(d3d::k_lines_deep),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(d3d::k_lines_deep, dim3((unsigned)c->HW), dim3(1024), lds, c->stream, A,
                           params, (const uint8_t *)c->mask, out, convolved);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    switch (pick_nt(c->HL)) {
        case 256: return launch_lines_nt<256>(c, out, convolved, params);
        case 512: return launch_lines_nt<512>(c, out, convolved, params);
        default: return launch_lines_nt<1024>(c, out, convolved, params);
    }
}

template <int NT>
int launch_spectral_nt(d3d_ctx *c, const double *in, double *out) {
    d3d::SpectralArgs A = spectral_args(c);
    const int G = NT / c->HL;
    const unsigned grid = (unsigned)((c->HW + G - 1) / G);
    const size_t lds = (size_t)G * c->N * sizeof(double);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(d3d::k_spectral<NT>), dim3(grid), dim3(NT), lds, c->stream,
                       A, in, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_spectral(d3d_ctx *c, const double *in, double *out) {
    // taps within +-8 channels, any depth that is not served by the one-wavefront dense form:
    // 128-channel blocks, one wavefront each (streaming)
    if (c->lsf_dense_any && c->spectral_blocks && !(c->lsf_fusable && c->spectral_dense)) {
        const int nzb = (c->Dp + 127) / 128;
        const long nwaves = (long)c->HW * nzb;
        hipLaunchKernelGGL(d3d::k_spectral_blocks, dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0,
                           c->stream, c->D, c->Dp, c->N, nzb, nwaves, (const double *)c->lsf_dense, in,
                           out);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (c->deep) {
        d3d::SpectralArgs A = spectral_args(c);
        const size_t lds = (size_t)c->N * sizeof(double);
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(d3d::k_spectral_deep),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(d3d::k_spectral_deep, dim3((unsigned)c->HW), dim3(1024), lds, c->stream, A,
                           in, out);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (c->lsf_fusable && c->spectral_dense) {
        // dense +-LSF_RL taps, spectrum within one wavefront: streaming form
        const int NT = 256, G = NT / c->HL;
        const unsigned grid = (unsigned)((c->HW + G - 1) / G);
#ifdef D3D_EXPERIMENTS
        if (c->spectral_shfl && c->HL == 64) {  // neighbours by wavefront shuffles, no LDS
            hipLaunchKernelGGL(HIP_KERNEL_NAME(d3d::k_spectral_shfl<256>), dim3(grid), dim3(NT), 0,
                               c->stream, c->Dp, c->HW, (const double *)c->lsf_dense, in, out);
            HIP_TRY(hipGetLastError());
            return 0;
        }
#endif
        const size_t lds = (size_t)G * (c->Dp + 2 * d3d::LSF_RL) * sizeof(double);
        hipLaunchKernelGGL(HIP_KERNEL_NAME(d3d::k_spectral_dense<256>), dim3(grid), dim3(NT), lds,
                           c->stream, c->Dp, c->HL, c->HW, (const double *)c->lsf_dense, in, out);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    switch (pick_nt(c->HL)) {
        case 256: return launch_spectral_nt<256>(c, in, out);
        case 512: return launch_spectral_nt<512>(c, in, out);
        default: return launch_spectral_nt<1024>(c, in, out);
    }
}

template <int NT, int FW>
int launch_spatial_fw(d3d_ctx *c, const d3d::SpatialArgs &A, const double *in, double *out) {
    constexpr int TX = 8;
    const int S = NT / c->HL;
    const long strips = (long)c->H * ((c->W + TX - 1) / TX);
    const unsigned grid = (unsigned)((strips + S - 1) / S);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(d3d::k_spatial<NT, FW, TX>), dim3(grid), dim3(NT), 0,
                       c->stream, A, in, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int NT, int FS, int TX, bool SYMX, bool UNI, bool FUSE, bool SYMY>
int launch_march(d3d_ctx *c, const d3d::SpatialArgs &A, const double *in, double *out) {
    const int S = NT / c->HL;
    const int HY = c->march_hy;
    const long items = (long)((c->W + TX - 1) / TX) * ((c->H + HY - 1) / HY);
    const unsigned grid = (unsigned)((items + S - 1) / S);
    const size_t lds =
        FUSE ? (size_t)S * TX * (c->Dp + 2 * d3d::LSF_RL) * sizeof(double) : 0;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(d3d::k_spatial_march<NT, FS, TX, SYMX, UNI, FUSE, SYMY>),
                       dim3(grid), dim3(NT), lds, c->stream, A, in, out, HY);
    HIP_TRY(hipGetLastError());
    return 0;
}

#ifdef D3D_EXPERIMENTS
// Diagnostic build (D3D_STAMP=1): the xy-symmetric march kernel with in-kernel
// s_memtime stamps; prints the per-phase cycle shares of a march step to stderr.
template <int NT, int FS, int TX>
int launch_march_stamped(d3d_ctx *c, d3d::SpatialArgs A, const double *in, double *out) {
    const int S = NT / c->HL;
    const int HY = c->march_hy;
    const long items = (long)((c->W + TX - 1) / TX) * ((c->H + HY - 1) / HY);
    const unsigned grid = (unsigned)((items + S - 1) / S);
    const size_t nw = (size_t)grid * (NT / 64);
    unsigned long long *dbg = nullptr;
    HIP_TRY(hipMalloc(&dbg, nw * 8 * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(dbg, 0, nw * 8 * sizeof(unsigned long long), c->stream));
    A.dbg = dbg;
    hipLaunchKernelGGL(HIP_KERNEL_NAME(d3d::k_spatial_march<NT, FS, TX, true, true, false, true, true>),
                       dim3(grid), dim3(NT), 0, c->stream, A, in, out, HY);
    HIP_TRY(hipGetLastError());
    std::vector<unsigned long long> h(nw * 8);
    HIP_TRY(hipMemcpyAsync(h.data(), dbg, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost,
                           c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    (void)hipFree(dbg);
    double sum[5] = {0, 0, 0, 0, 0}, steps = 0;
    unsigned long long tmin = ~0ULL, tmax = 0;
    size_t live = 0;
    for (size_t w = 0; w < nw; ++w) {
        if (!h[w * 8 + 5]) continue;
        ++live;
        for (int k = 0; k < 5; ++k) sum[k] += (double)h[w * 8 + k];
        steps += (double)h[w * 8 + 5];
        if (h[w * 8 + 6] < tmin) tmin = h[w * 8 + 6];
        if (h[w * 8 + 7] > tmax) tmax = h[w * 8 + 7];
    }
    fprintf(stderr,
            "[d3d stamp] waves %zu steps/wave %.1f | cycles per step: issue %.0f wait %.0f math %.0f "
            "tail %.0f | wave lifetime %.0f cyc | kernel span %.0f cyc\n",
            live, steps / live, sum[0] / steps, sum[1] / steps, sum[2] / steps, sum[3] / steps,
            sum[4] / live, (double)(tmax - tmin));
    return 0;
}

#endif  // D3D_EXPERIMENTS

// One pass for LSF x outer-product FSF (A.lsf_dense set; a strip within a wavefront).
template <int NT, int FS>
int launch_sep_lsf(d3d_ctx *c, const d3d::SpatialArgs &A, const double *in, double *out) {
    constexpr int TX = (FS >= 9 ? 3 : 4);
    const int S = NT / c->HL;
    const int HY = c->march_hy;
    const long items = (long)((c->W + TX - 1) / TX) * ((c->H + HY - 1) / HY);
    const unsigned grid = (unsigned)((items + S - 1) / S);
    const size_t lds = (size_t)S * TX * (c->Dp + 2 * d3d::LSF_RL) * sizeof(double);
    if ((c->HL % 64) == 0)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(d3d::k_spatial_sep_lsf<NT, FS, TX, true>), dim3(grid),
                           dim3(NT), lds, c->stream, A, in, out, HY);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(d3d::k_spatial_sep_lsf<NT, FS, TX, false>), dim3(grid),
                           dim3(NT), lds, c->stream, A, in, out, HY);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int NT, int FS>
int launch_sep(d3d_ctx *c, const d3d::SpatialArgs &A, const double *in, double *out) {
    if (A.lsf_dense) return launch_sep_lsf<NT, FS>(c, A, in, out);
    constexpr int TX = (FS >= 9 ? 3 : 4);
    const int S = NT / c->HL;
    const int HY = c->march_hy;
    const long items = (long)((c->W + TX - 1) / TX) * ((c->H + HY - 1) / HY);
    const unsigned grid = (unsigned)((items + S - 1) / S);
    if ((c->HL % 64) == 0)
        hipLaunchKernelGGL(HIP_KERNEL_NAME(d3d::k_spatial_sep<NT, FS, TX, true>), dim3(grid),
                           dim3(NT), 0, c->stream, A, in, out, HY);
    else
        hipLaunchKernelGGL(HIP_KERNEL_NAME(d3d::k_spatial_sep<NT, FS, TX, false>), dim3(grid),
                           dim3(NT), 0, c->stream, A, in, out, HY);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int NT, int FS, bool FUSE>
int launch_march_fs(d3d_ctx *c, const d3d::SpatialArgs &A, const double *in, double *out) {
    if (A.sep_uv && (!FUSE || c->sep_fuse)) return launch_sep<NT, FS>(c, A, in, out);
    constexpr int TX = (FS >= 9 ? 3 : 4);
    const bool uni = (c->HL % 64) == 0;  // a wavefront never straddles two strips
    const bool symx = c->march_mode >= 2 && c->fsf_symx;
    const bool symxy = symx && c->fsf_symy && c->march_mode != 3;  // mode 3: x symmetry only
#ifdef D3D_EXPERIMENTS
    if (symxy && !FUSE && c->march_one > 0 && c->Dp % 64 == 0 && NT % c->Dp == 0) {
        // one channel per lane: 3 (TX = 3) or 4 (TX = 2) wavefronts per SIMD
        const int HY = c->march_hy;
        const int S = NT / c->Dp;
        if (c->march_one == 3) {
            const long items = (long)((c->W + 2) / 3) * ((c->H + HY - 1) / HY);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(d3d::k_spatial_march1<NT, FS, 3, 3>),
                               dim3((unsigned)((items + S - 1) / S)), dim3(NT), 0, c->stream, A, in,
                               out, HY);
        } else {
            const long items = (long)((c->W + 1) / 2) * ((c->H + HY - 1) / HY);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(d3d::k_spatial_march1<NT, FS, 2, 4>),
                               dim3((unsigned)((items + S - 1) / S)), dim3(NT), 0, c->stream, A, in,
                               out, HY);
        }
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (uni && symxy && !FUSE && c->march_pf > 0) {
        // software-pipelined variant (next-row loads interleaved with the FMAs)
        const int HY = c->march_hy;
        const int S = NT / c->HL;
        if (c->march_pf == 3) {
            const long items = (long)((c->W + 2) / 3) * ((c->H + HY - 1) / HY);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(d3d::k_spatial_march_pf<NT, FS, 3>),
                               dim3((unsigned)((items + S - 1) / S)), dim3(NT), 0, c->stream, A, in,
                               out, HY);
        } else {
            const long items = (long)((c->W + 1) / 2) * ((c->H + HY - 1) / HY);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(d3d::k_spatial_march_pf<NT, FS, 2>),
                               dim3((unsigned)((items + S - 1) / S)), dim3(NT), 0, c->stream, A, in,
                               out, HY);
        }
        HIP_TRY(hipGetLastError());
        return 0;
    }
#endif
    if (uni) {
#ifdef D3D_EXPERIMENTS
        if constexpr (FS == 11 && !FUSE && NT == 256) {
            if (symxy && c->march_stamp) return launch_march_stamped<NT, FS, TX>(c, A, in, out);
        }
#endif
        if (symxy) return launch_march<NT, FS, TX, true, true, FUSE, true>(c, A, in, out);
        if (symx) return launch_march<NT, FS, TX, true, true, FUSE, false>(c, A, in, out);
        return launch_march<NT, FS, TX, false, true, FUSE, false>(c, A, in, out);
    }
    if (symxy) return launch_march<NT, FS, TX, true, false, FUSE, true>(c, A, in, out);
    if (symx) return launch_march<NT, FS, TX, true, false, FUSE, false>(c, A, in, out);
    return launch_march<NT, FS, TX, false, false, FUSE, false>(c, A, in, out);
}

template <int NT, bool FUSE>
int launch_march_any(d3d_ctx *c, const d3d::SpatialArgs &A, const double *in, double *out,
                     bool *done) {
    *done = true;
    switch (c->fw) {
        case 3: return launch_march_fs<NT, 3, FUSE>(c, A, in, out);
        case 5: return launch_march_fs<NT, 5, FUSE>(c, A, in, out);
        case 7: return launch_march_fs<NT, 7, FUSE>(c, A, in, out);
        case 9: return launch_march_fs<NT, 9, FUSE>(c, A, in, out);
        case 11: return launch_march_fs<NT, 11, FUSE>(c, A, in, out);
        case 13: return launch_march_fs<NT, 13, FUSE>(c, A, in, out);
        case 15: return launch_march_fs<NT, 15, FUSE>(c, A, in, out);
        default: break;
    }
    *done = false;
    return 0;
}

template <int NT>
int launch_spatial_nt(d3d_ctx *c, const double *in, double *out, const double *data,
                      bool fuse_lsf) {
    d3d::SpatialArgs A;
    A.Dp = c->Dp;
    A.HL = c->HL;
    A.H = c->H;
    A.W = c->W;
    A.fh = c->fh;
    A.fw = c->fw;
    A.fsf = c->fsf;
    A.data = data;
    A.lsf_dense = nullptr;
    A.sep_uv = (c->fsf_sep && (!fuse_lsf || c->sep_fuse)) ? c->sep_uv : nullptr;
    A.xcd_remap = c->xcd_remap;
    A.alt_dir = c->alt_dir;
    A.dbg = nullptr;
    A.stagger = c->stagger;
    // (the march kernels are built for 256-thread groups only: D <= 512; deeper
    // cubes use the tile kernel below)
    if constexpr (NT == 256)
    if (c->march_mode > 0 && c->fh == c->fw) {
        bool done = false;
        int rc;
        if (fuse_lsf && A.sep_uv && c->sep_fuse) {  // LSF x outer-product FSF in one pass
            A.lsf_dense = c->lsf_dense;
            rc = launch_march_any<NT, false>(c, A, in, out, &done);
            if (done) return rc;
            A.lsf_dense = nullptr;
        }
#ifdef D3D_EXPERIMENTS
        if (fuse_lsf) {
            A.lsf_dense = c->lsf_dense;
            rc = launch_march_any<NT, true>(c, A, in, out, &done);
        } else
#endif
        {
            rc = launch_march_any<NT, false>(c, A, in, out, &done);
        }
        if (done) return rc;
        A.lsf_dense = nullptr;
    }
    if (fuse_lsf) return fail(D3D_ERR_STATE, "internal: fused LSF requested without march kernel");
    switch (c->fw) {
        case 1: return launch_spatial_fw<NT, 1>(c, A, in, out);
        case 3: return launch_spatial_fw<NT, 3>(c, A, in, out);
        case 5: return launch_spatial_fw<NT, 5>(c, A, in, out);
        case 7: return launch_spatial_fw<NT, 7>(c, A, in, out);
        case 9: return launch_spatial_fw<NT, 9>(c, A, in, out);
        case 11: return launch_spatial_fw<NT, 11>(c, A, in, out);
        case 13: return launch_spatial_fw<NT, 13>(c, A, in, out);
        case 15: return launch_spatial_fw<NT, 15>(c, A, in, out);
        default: break;
    }
    const int S = NT / c->HL;
    const unsigned grid = (unsigned)((c->HW + S - 1) / S);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(d3d::k_spatial_generic<NT>), dim3(grid), dim3(NT), 0,
                       c->stream, A, in, out);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- one-pass kernel k_conv_rows (d3d_conv.h) ---------------------------------------
// Usable for a 128-channel spectrum (one wavefront per column) and a square FSF with
// both mirror symmetries; with_lsf additionally needs the dense power-of-two LSF form.
bool conv_rows_usable(const d3d_ctx *c, bool with_lsf) {
    if (!c->conv_rows) return false;
    if (!(c->fsf_symx && c->fsf_symy && c->fh == c->fw)) return false;
    // (the LSF epilogue is built for mirror-symmetric taps only -- every Gaussian / MUSE-like
    // LSF; an asymmetric one gets its own pass)
    if (with_lsf && !(c->ntaps > 0 && c->lsf_dense_ok && c->N == c->D && c->lsf_dense_sym)) return false;
    // (15 x 15 -- the FSF of the reference's own science fixture, tests/read_mat.py:28-36 --
    // keeps its taps in scalar registers only in the radial and outer-product forms: 36 / 16
    // values; the general quadrant form would need 64)
    const bool fs15_ok = c->fsf_symt || (c->fsf_sep && c->march_mode > 0);
    if (c->Dp == 64 || c->Dp == 32)  // several spectra per wavefront: the BASELINE footprints,
        return (c->fw == 9 || c->fw == 11) &&
               (c->fsf_symt || (c->fsf_sep && c->march_mode > 0));  // radial or outer-product FSFs
    // every other depth: z-blocks of 128 channels (the last -- below 128 channels the only --
    // one ragged), FSF only: the LSF couples the blocks, and at a depth that is not a power of
    // two it wraps partially (lib/convolution.py:123-160)
    if (c->Dp != d3d::CONV_DP)
        return !with_lsf && c->conv_zb &&
               (c->fw == 9 || c->fw == 11 || c->fw == 13 || (c->fw == 15 && fs15_ok));
    switch (c->fw) {
        case 3: case 5: case 7: case 9: case 11: case 13: return true;
        case 15: return fs15_ok;
        default: return false;
    }
}

template <int FS, bool LSF, bool LSYM, bool RESID, int TSYM, int DPS = d3d::CONV_DP, bool ZB = false>
int launch_conv_rows_t(d3d_ctx *c, const double *in, double *out, const double *data) {
    constexpr int NW = 15;
    constexpr int NWC = d3d::ConvGeo<FS, NW, DPS>::NWC;  // output columns per workgroup
    d3d::ConvRowsArgs A;
    A.H = c->H;
    A.W = c->W;
    A.Dp = c->Dp;
    A.ngx = (c->W + NWC - 1) / NWC;
    const int nzb = ZB ? (c->Dp + d3d::CONV_DP - 1) / d3d::CONV_DP : 1;  // z-blocks of 128 channels
    // one workgroup per CU (1024 threads, ~93 KB of LDS): as many row strips as fill the
    // chip in ONE round
    const int cus = c->flow_grid > 0 ? c->flow_grid / 4 : 256;
    int ngy = std::max(1, cus / (A.ngx * nzb));
    ngy = std::min(ngy, c->H);
    A.HY = (c->H + ngy - 1) / ngy;
    if (c->conv_hy_opt >= 1) A.HY = c->conv_hy_opt;
    A.ngy = (c->H + A.HY - 1) / A.HY;
    A.xcd_remap = 1;
    auto kern = d3d::k_conv_rows<FS, NW, LSF, LSYM, RESID, TSYM, DPS, ZB>;
    constexpr size_t lds = d3d::conv_rows_lds_bytes<FS, NW, DPS>();
    // > 64 KB of dynamic LDS has to be allowed per function AND per device: set on every
    // launch (a host-side call of a few microseconds; this kernel is not in the MH loop)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)(A.ngx * A.ngy * nzb)), dim3((NW + 1) * 64), lds, c->stream, A,
                       in, out, (const double *)(TSYM == 2 ? c->fsf_quad_sep : c->fsf_quad),
                       (const double *)c->lsf_dense, data);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int FS, int TSYM, int DPS = d3d::CONV_DP>
int launch_conv_rows_ts(d3d_ctx *c, const double *in, double *out, const double *data, bool lsf) {
    if (lsf) {  // (conv_rows_usable: symmetric LSF taps)
        if (data) return launch_conv_rows_t<FS, true, true, true, TSYM, DPS>(c, in, out, data);
        return launch_conv_rows_t<FS, true, true, false, TSYM, DPS>(c, in, out, data);
    }
    if (data) return launch_conv_rows_t<FS, false, false, true, TSYM, DPS>(c, in, out, data);
    return launch_conv_rows_t<FS, false, false, false, TSYM, DPS>(c, in, out, data);
}

template <int FS, int DPS = d3d::CONV_DP>
int launch_conv_rows_fs(d3d_ctx *c, const double *in, double *out, const double *data, bool lsf) {
    if (c->fsf_sep && c->march_mode > 0) return launch_conv_rows_ts<FS, 2, DPS>(c, in, out, data, lsf);
    if (c->fsf_symt) return launch_conv_rows_ts<FS, 1, DPS>(c, in, out, data, lsf);
    if constexpr (DPS == d3d::CONV_DP && FS < 15) return launch_conv_rows_ts<FS, 0, DPS>(c, in, out, data, lsf);
    return fail(D3D_ERR_STATE, "internal: no one-pass kernel for this FSF at %d channels", c->Dp);
}

// any depth above 128: z-blocks, FSF only
template <int FS>
int launch_conv_rows_zb(d3d_ctx *c, const double *in, double *out, const double *data) {
    constexpr int DP = d3d::CONV_DP;
    if (c->fsf_sep && c->march_mode > 0)
        return data ? launch_conv_rows_t<FS, false, false, true, 2, DP, true>(c, in, out, data)
                    : launch_conv_rows_t<FS, false, false, false, 2, DP, true>(c, in, out, data);
    if (c->fsf_symt)
        return data ? launch_conv_rows_t<FS, false, false, true, 1, DP, true>(c, in, out, data)
                    : launch_conv_rows_t<FS, false, false, false, 1, DP, true>(c, in, out, data);
    if constexpr (FS < 15)
        return data ? launch_conv_rows_t<FS, false, false, true, 0, DP, true>(c, in, out, data)
                    : launch_conv_rows_t<FS, false, false, false, 0, DP, true>(c, in, out, data);
    return fail(D3D_ERR_STATE, "internal: no one-pass kernel for a general 15 x 15 FSF");
}

int launch_conv_rows(d3d_ctx *c, const double *in, double *out, const double *data, bool lsf) {
    if (c->Dp != 64 && c->Dp != 32 && c->Dp != d3d::CONV_DP) {
        if (lsf) return fail(D3D_ERR_STATE, "internal: one-pass LSF requested at %d channels", c->Dp);
        switch (c->fw) {
            case 9: return launch_conv_rows_zb<9>(c, in, out, data);
            case 11: return launch_conv_rows_zb<11>(c, in, out, data);
            case 13: return launch_conv_rows_zb<13>(c, in, out, data);
            default: return launch_conv_rows_zb<15>(c, in, out, data);
        }
    }
    if (c->Dp == 64)
        return c->fw == 9 ? launch_conv_rows_fs<9, 64>(c, in, out, data, lsf)
                          : launch_conv_rows_fs<11, 64>(c, in, out, data, lsf);
    if (c->Dp == 32)
        return c->fw == 9 ? launch_conv_rows_fs<9, 32>(c, in, out, data, lsf)
                          : launch_conv_rows_fs<11, 32>(c, in, out, data, lsf);
    switch (c->fw) {
        case 3: return launch_conv_rows_fs<3>(c, in, out, data, lsf);
        case 5: return launch_conv_rows_fs<5>(c, in, out, data, lsf);
        case 7: return launch_conv_rows_fs<7>(c, in, out, data, lsf);
        case 9: return launch_conv_rows_fs<9>(c, in, out, data, lsf);
        case 11: return launch_conv_rows_fs<11>(c, in, out, data, lsf);
        case 13: return launch_conv_rows_fs<13>(c, in, out, data, lsf);
        default: return launch_conv_rows_fs<15>(c, in, out, data, lsf);
    }
}

// True when the spatial pass can apply the LSF itself (fused epilogue).
bool can_fuse_lsf(const d3d_ctx *c) {
    if (c->deep) return false;
    // (an outer-product FSF honours D3D_SEP_FUSE=0: LSF in its own pass, for A/B tests)
    if (conv_rows_usable(c, true) && !(c->fsf_sep && c->march_mode > 0 && !c->sep_fuse)) return true;
    if (!c->lsf_fusable || c->march_mode <= 0 || c->fh != c->fw) return false;
    const bool sep = c->fsf_sep && c->sep_fuse;  // k_spatial_sep_lsf
#ifndef D3D_EXPERIMENTS
    // the fused epilogue of the 2-D march kernel is an experiment (register spills:
    // slower than the streaming LSF pass)
    if (!sep) return false;
#else
    if (!sep && !c->fuse_lsf) return false;
#endif
    switch (c->fw) {
        case 3: case 5: case 7: case 9: case 11: case 13: case 15: return true;
        default: return false;
    }
}

// out = FSF (*) in, or data - FSF (*) in when data != NULL.  in != out.
// fuse_lsf: also apply the LSF along z (only when can_fuse_lsf()).
int launch_spatial(d3d_ctx *c, const double *in, double *out, const double *data,
                   bool fuse_lsf) {
    if (conv_rows_usable(c, fuse_lsf)) return launch_conv_rows(c, in, out, data, fuse_lsf);
    if (c->deep) {
        if (fuse_lsf) return fail(D3D_ERR_STATE, "internal: fused LSF requested on a deep cube");
        d3d::SpatialArgs A = {};
        A.Dp = c->Dp;
        A.HL = c->HL;
        A.H = c->H;
        A.W = c->W;
        A.fh = c->fh;
        A.fw = c->fw;
        A.fsf = c->fsf;
        A.data = data;
        hipLaunchKernelGGL(d3d::k_spatial_deep, dim3((unsigned)c->HW, (unsigned)((c->HL + 255) / 256)),
                           dim3(256), 0, c->stream, A, in, out);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    int nt = pick_nt(c->HL);
    if (c->sp_nt_opt >= nt) nt = c->sp_nt_opt;
    switch (nt) {
        case 256: return launch_spatial_nt<256>(c, in, out, data, fuse_lsf);
        case 512: return launch_spatial_nt<512>(c, in, out, data, fuse_lsf);
        default: return launch_spatial_nt<1024>(c, in, out, data, fuse_lsf);
    }
}

// params -> SLOT_TMP0 (LSF lines) -> dst (sim, or residual when resid)
int forward_into(d3d_ctx *c, double *dst, bool resid) {
    if (resid) pend_clear(c);  // a fresh residual supersedes pending updates
    const double *data = resid ? c->slot[D3D_SLOT_DATA] : nullptr;
    // FSF and LSF commute: where the spatial kernel can apply the LSF in its epilogue the
    // lines are built raw (exp only) and the LSF costs no pass of its own
    if (c->ntaps > 0 && can_fuse_lsf(c)) {
        int rc = launch_lines(c, c->slot[D3D_SLOT_TMP0], 0);
        if (rc) return rc;
        return launch_spatial(c, c->slot[D3D_SLOT_TMP0], dst, data, true);
    }
    int rc = launch_lines(c, c->slot[D3D_SLOT_TMP0], 1);
    if (rc) return rc;
    return launch_spatial(c, c->slot[D3D_SLOT_TMP0], dst, data);
}

bool zmajor_ok(const d3d_ctx *c) {
    if (!c->zmajor) return false;
    if (!(c->fsf_symx && c->fsf_symy && c->fh == c->fw)) return false;
    if (c->ntaps > 0 && !c->lsf_dense_ok) return false;
    switch (c->fw) {
        case 3: case 5: case 7: case 9: case 11: case 13: case 15: return true;
        default: return false;
    }
}

template <int FS>
int launch_spatial_z(d3d_ctx *c, const double *in, double *out) {
    // rows per strip (option zmajor_hy; 0 = by shape).  A wavefront's march is a dependent chain
    // of HY + FS - 1 steps of ~ 0.75 us whatever else runs, so: the shortest strips whose
    // wavefronts are all resident at once (4 per SIMD: 4096), at least 4 rows --
    // 300x300x128: 50 rows, 116 against 134 us per convolution at 32; 64^3: 4 rows, 28.5 against
    // 40.7; 300x300x256: 100 rows, 290 against 317 (tools/zmajor_time.py).
    int HY = c->zmajor_hy;
    if (HY <= 0) {
        const long per_row_of_strips = (long)c->D * ((c->W + 63) / 64);
        const long nys = std::max<long>(1, 4096 / per_row_of_strips);
        HY = (int)std::max<long>(4, (c->H + nys - 1) / nys);
    }
    const long items = (long)c->D * ((c->H + HY - 1) / HY) * ((c->W + 63) / 64);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(d3d::k_spatial_z<FS>), dim3((unsigned)((items + 3) / 4)),
                       dim3(256), 0, c->stream, c->D, c->H, c->W, HY, (const double *)c->fsf, in,
                       out);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_zmajor_convolve(d3d_ctx *c) {
    // both passes in the reference layout, lanes along x: no layout change
    const double *src = c->stage;
    if (c->ntaps > 0) {
        hipLaunchKernelGGL(d3d::k_spectral_z, dim3((unsigned)((c->HW + 255) / 256)), dim3(256), 0,
                           c->stream, c->D, c->HW, (const double *)c->lsf_dense,
                           (const double *)c->stage, c->stage2);
        HIP_TRY(hipGetLastError());
        src = c->stage2;
    }
    double *dst = (src == c->stage) ? c->stage2 : c->stage;
    int rc;
    switch (c->fw) {
        case 3: rc = launch_spatial_z<3>(c, src, dst); break;
        case 5: rc = launch_spatial_z<5>(c, src, dst); break;
        case 7: rc = launch_spatial_z<7>(c, src, dst); break;
        case 9: rc = launch_spatial_z<9>(c, src, dst); break;
        case 11: rc = launch_spatial_z<11>(c, src, dst); break;
        case 13: rc = launch_spatial_z<13>(c, src, dst); break;
        default: rc = launch_spatial_z<15>(c, src, dst); break;
    }
    if (rc) return rc;
    if (dst != c->stage)
        HIP_TRY(hipMemcpyAsync(c->stage, dst, (size_t)c->D * c->HW * sizeof(double),
                               hipMemcpyDeviceToDevice, c->stream));
    return D3D_OK;
}

}  // namespace d3dh
