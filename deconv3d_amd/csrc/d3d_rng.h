// Counter-based RNG and truncated-normal sampler of the MH-within-Gibbs kernel.
//
// The reference draws from the unseeded global numpy RNG (lib/run.py:313,435,
// 578) and from Chopin's table sampler (lib/rtnorm.py:95-224, GPL tables).
// Neither is reproduced: the device uses Philox4x32-10 keyed by
// (seed, global spaxel index, sweep, block) so that an update's random numbers
// do not depend on launch geometry, tiling or scan order, and an own
// truncated-normal sampler with the same distribution as rtnorm.  The oracle
// (oracle/deconv3d_oracle.py: philox_pair, truncated_normal) restates both bit
// for bit / formula for formula.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace d3d {

struct U2 {
    double x, y;
};

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
        const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += W0;
        k1 += W1;
    }
    out[0] = c0;
    out[1] = c1;
    out[2] = c2;
    out[3] = c3;
}

// (0,1) double from 64 random bits: ((u >> 12) + 0.5) * 2^-52, exact in fp64,
// in [2^-53, 1 - 2^-53] (never 0 or 1).
__device__ __forceinline__ double u64_to_unit(uint64_t u) {
    return ((double)(u >> 12) + 0.5) * (1.0 / 4503599627370496.0);
}

// Two uniforms of block `block` of the update (spaxel, sweep).
__device__ __forceinline__ U2 philox_pair(uint64_t seed, uint32_t spaxel, uint32_t sweep,
                                          uint32_t block) {
    uint32_t r[4];
    philox4x32_10(spaxel, sweep, block, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    U2 u;
    u.x = u64_to_unit(((uint64_t)r[1] << 32) | r[0]);
    u.y = u64_to_unit(((uint64_t)r[3] << 32) | r[2]);
    return u;
}

// Philox block layout of one spaxel update (oracle: BLK_*).
constexpr uint32_t BLK_JUMP_AC = 0;  // (u_a, u_c)
constexpr uint32_t BLK_JUMP_W = 1;   // (u_w, u_accept)
constexpr uint32_t BLK_GIBBS = 2;    // truncated normal: 2, 3, ...

constexpr double TN_TAIL = 6.0;
constexpr double SQRT2 = 1.4142135623730951;

// N(0,1) truncated to [alpha, beta]; consumes blocks from *blk upwards.  u0 is
// the pair of block *blk, drawn by the caller ahead of time (it depends on
// nothing the window pass produces).  WAVE: every lane of the wavefront calls
// with the same arguments; the two CDF evaluations then run side by side in the
// two halves of the wavefront instead of one after the other -- same functions,
// same arguments, same bits.
template <bool WAVE>
__device__ inline double truncated_standard_normal(double alpha, double beta, U2 u0, uint64_t seed,
                                                   uint32_t spaxel, uint32_t sweep,
                                                   uint32_t *blk) {
    double sign = 1.0;
    if (beta <= 0.0) {  // mirror so that the interval reaches into x > 0
        const double t = -alpha;
        alpha = -beta;
        beta = t;
        sign = -1.0;
    }
    double z;
    if (alpha >= TN_TAIL) {
        // Robert (1995): translated exponential proposal, rate lam
        const double lam = 0.5 * (alpha + sqrt(alpha * alpha + 4.0));
        z = alpha;
        for (int it = 0; it < 1000; ++it) {
            const U2 u = (it == 0) ? u0 : philox_pair(seed, spaxel, sweep, *blk);
            ++*blk;
            const double zz = alpha - log(u.x) / lam;
            if (zz <= beta && log(u.y) <= -0.5 * (zz - lam) * (zz - lam)) {
                z = zz;
                break;
            }
        }
        return sign * z;
    }
    ++*blk;
    const bool upper = WAVE && (__lane_id() & 32);
    const double x = upper ? beta : alpha;
    if (alpha > 0.0) {
        double qa = 0.5 * erfc(x / SQRT2), qb;
        if (WAVE) {
            qb = __shfl(qa, 32);
            qa = __shfl(qa, 0);
        } else {
            qb = 0.5 * erfc(beta / SQRT2);
        }
        const double q = qa - u0.x * (qa - qb);
        z = SQRT2 * erfcinv(2.0 * q);
    } else {
        double pa = normcdf(x), pb;
        if (WAVE) {
            pb = __shfl(pa, 32);
            pa = __shfl(pa, 0);
        } else {
            pb = normcdf(beta);
        }
        z = normcdfinv(pa + u0.x * (pb - pa));
    }
    z = fmin(fmax(z, alpha), beta);
    return sign * z;
}

// TN(lo, hi; mu, sigma): distribution of rtnorm(lo, hi, mu, sigma), lib/rtnorm.py:21-92.
template <bool WAVE>
__device__ inline double truncated_normal(double lo, double hi, double mu, double sigma, U2 u0,
                                          uint64_t seed, uint32_t spaxel, uint32_t sweep,
                                          uint32_t *blk) {
    double alpha, beta;
    if (WAVE) {  // the two standardised bounds side by side, as the two CDFs below
        const double x = (((__lane_id() & 32) ? hi : lo) - mu) / sigma;
        alpha = __shfl(x, 0);
        beta = __shfl(x, 32);
    } else {
        alpha = (lo - mu) / sigma;
        beta = (hi - mu) / sigma;
    }
    const double z = truncated_standard_normal<WAVE>(alpha, beta, u0, seed, spaxel, sweep, blk);
    return fmin(fmax(mu + sigma * z, lo), hi);
}

}  // namespace d3d
