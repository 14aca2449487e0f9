// k_mh_small -- the colour launch that does NOT fill the chip (round 4).  gfx950 only.
//
// lib/run.py:367-519 for one colour class, deferred write-back with ONE pending layer: the
// arithmetic of k_mh_ws<NS, UV, U, 1, ...> (d3d_kernels.h) -- same thread <-> (position group,
// z-pair) mapping, same summation order, same decision code: the chain is bit-identical to it
// -- rebuilt around what the phase stamps of such a launch show (tools/mh_tail.py,
// profiles/r04_small_launch.txt; 64x64x64: 49 windows on 49 compute units, one wavefront
// per SIMD):
//
//   * the window pass was neither bandwidth- nor memory-latency-bound -- with the residual
//     stores AND the 1/variance loads switched off it still took 5.3 of 5.9 us, and with
//     every load of a thread requested at once it took as long -- but bound by INSTRUCTION
//     ISSUE: ~115 vector instructions per window position and wavefront (position -> voxel
//     column, covering spaxel of the pending colour, its tap, 64-bit addresses; then three
//     dependent LDS round trips in k_mh_ws), 16 positions per thread, and nothing to hide a
//     4-cycle issue slot behind when a workgroup is alone on its compute unit.  But the
//     geometry of a window is the same for every window of a launch: the pending colour's
//     lattice has the period of the launch's own, so "which pending spaxel covers position p,
//     with which tap" depends on the two colour classes only.  The host precomputes it once
//     per context, for every relative offset of the two lattices (MHPos tables, d3d_set_taps);
//     a workgroup copies its launch's table into LDS and adds only what differs per window:
//     the base voxel and four small validity masks (rows / columns inside the launch's domain,
//     the <= 2 x 2 covering spaxels inside the cube).  ~40 instructions per position; 32-bit
//     byte offsets from a scalar base instead of 64-bit address arithmetic;
//   * what is left then IS memory latency (~1.2 us per round trip right after a kernel
//     boundary: cold L2s), so a thread requests its whole share of the window -- up to U = 16
//     positions, 32 loads -- at once, before the setup's barrier;
//   * no prepare wavefront: the LSF-convolved unit lines of the current and the proposed
//     (c, w) come from the sweep's LINE TABLE (k_mh_line_table: one chip-filling launch per
//     sweep computes them for every spaxel with the very code the prepare wavefront ran --
//     a spaxel is visited once per sweep, so both are known at its start, like the proposal,
//     lib/run.py:369-395); a streaming thread loads its channel's two values at entry;
//   * the tail runs on the wavefronts that hold channels only (one at <= 64 channels, two at
//     128): the others leave after the barrier that publishes the group sums; with one
//     channel wavefront the seven totals never touch LDS and there is no further barrier;
//     with more, every channel wavefront takes the (deterministic) decision itself instead
//     of waiting for a verdict.
#pragma once
#include <hip/hip_runtime.h>

namespace d3d {

// The lattice point of class c (period per) whose window (half width hw) covers coordinate q:
// covering_coord without the range test.
__host__ __device__ inline int mh_raw_cover(int q, int c, int per, int hw) {
    int m = (q - c) % per;
    if (m < 0) m += per;
    int s = q - m;
    if (q - s > hw) s += per;
    return s;
}

// A position of the window as the launch's table holds it (32 bytes, one double4):
//   .x  f  = fsf[p], this launch's own tap            .y  fp = fsf[tap] of the pending update
//   .z  the bits of two 32-bit words: rel = spaxel offset of the voxel column from the window's
//       centre, (dy - fhh) W + (dx - fhw); pk = window row | column << 8 | staged G row
//       (2 hy + hx) << 16 | a pending tap applies << 18
// One table per relative offset (oy, ox) of the pending colour's lattice to the launch's own
// (row (oy + fhh) fw + ox + fhw of MHArgs::ptab), and one, row fh fw, for "nothing pending".
__host__ __device__ inline unsigned mh_pos_pack(int dy, int dx, int sel, int has) {
    return (unsigned)dy | ((unsigned)dx << 8) | ((unsigned)sel << 16) | ((unsigned)has << 18);
}

// One round of a streaming thread: U window positions, their loads in flight.
template <int U>
struct MHRound {
    unsigned off[U];  // byte offset of this thread's z-pair of the voxel column in SLOT_ERR / SLOT_IVAR
    // bit 0: inside the launch's domain; per pending layer j: bit 1 + 3 j: its update applies,
    // bits 2 + 3 j .. 3 + 3 j: which of its staged G rows; bits 8..: the window position (its
    // table entries hold the taps)
    unsigned fl[U];
    double2 e[U], v[U];
};

__host__ __device__ inline size_t mh_small_lds_doubles(int NS, int HL, int Dp, int npos, int M) {
    const int G = NS / HL;
    // per layer: position table (32 B per position), 4 staged G rows | group partial sums | wave sums, verdict
    return (size_t)M * (4 * (size_t)npos + 4 * (size_t)Dp) + (size_t)G * 3 * Dp + 8 * 8 + 8;
}

// K: registers per thread and layer that stage the <= 4 pending G rows (4 Dp <= K NS).
// BATCH: the launch holds the windows of R independent chains of one geometry, chain-major
// (d3d_mh_sweeps_batch: Run(..., chains=R) on a cube whose joint launch still does not fill the
// chip); the chain's cubes, parameters, bounds, random stream and sweep tables replace the
// arguments' -- work list, taps, position tables and pending-layer geometry are common.
// M: pending layers the kernel can apply (1; 2 for the chip-filling form, see k_mh_ws "several
// pending layers": the residual is stored by the launch that finds M layers pending).
// FULL: the launch fills the chip (several workgroups per compute unit): ONE wavefront takes the
// decision -- a second deciding wavefront per window takes issue slots from the neighbours'
// window passes there (measured in k_mh_ws: +0.8 us per launch) -- and few positions in flight.
// NTV (FULL only): the cache policy of a context beyond the Infinity Cache (k_mh_ws: 1/variance
// loaded non-temporally, the residual stored write-through through a raw buffer of the window).
template <int NS, bool UV, int U, int K, bool BATCH = false, int M = 1, bool FULL = false, bool NTV = false>
__global__ __launch_bounds__(NS) void k_mh_small(MHArgs P, uint32_t sweep) {
    static_assert(M == 1 || M == 2, "one or two pending layers");
    extern __shared__ double smem[];
    const int tid = threadIdx.x;
    const int HL = P.HL, Dp = P.Dp, npos = P.npos;
    const int G = NS / HL;
    const int g = tid / HL, zl = tid - g * HL;
    const bool active = g < G;
    const int fhh = (P.fh - 1) / 2, fhw = (P.fw - 1) / 2;
    double *s_tab = smem;  // [M][npos] MHPos
    double *s_gp = s_tab + (size_t)M * 4 * npos;  // [M][4][Dp]
    double *s_red = s_gp + (size_t)M * 4 * Dp;
    double *s_sum = s_red + (size_t)G * 3 * Dp;
    D3D_MH_STAMP(blockIdx.x, 0, 0);
#ifdef D3D_EXPERIMENTS
    if (P.stamp && threadIdx.x == 0)
        P.stamp[(long)blockIdx.x * 8 + 5] =
            (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) |
            ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 32);
#endif

    // ---- loads that do not depend on the work-list entry: this launch's position tables -------
    const double4 *T[M];
    double4 t0[M];
#pragma unroll
    for (int j = 0; j < M; ++j) {
        T[j] = reinterpret_cast<const double4 *>(P.ptab) + (size_t)P.ptab_row[j] * npos;
        t0[j] = make_double4(0, 0, 0, 0);
        if (tid < npos) t0[j] = T[j][tid];
    }
    const int per_thread = (npos + G - 1) / G;
    const int rounds = (per_thread + U - 1) / U;
    auto pos_of = [&](int r, int u) {  // window position of (round, slot), npos = none
        const int pw = g + (r * U + u) * G;
        return pw < npos ? (P.rev ? npos - 1 - pw : pw) : npos;
    };
    // (round 0's table entries straight from memory: the LDS copy is not there yet)
    double pe0[M][U];
#pragma unroll
    for (int j = 0; j < M; ++j)
#pragma unroll
        for (int u = 0; u < U; ++u)
            pe0[j][u] = reinterpret_cast<const double *>(T[j])[4 * (size_t)min(active ? pos_of(0, u) : npos, npos - 1) + 2];

    int blk = blockIdx.x;
    if constexpr (BATCH) {
        const int r = blockIdx.x / P.b_items;
        blk = blockIdx.x - r * P.b_items;
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
        for (int j = 0; j < M; ++j) P.lay_G[j] = B.gbuf[P.b_lay_g[j]];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            P.min_b[k] = B.min_b[k];
            P.max_b[k] = B.max_b[k];
            P.amp[k] = B.amp[k];
        }
        P.ra = B.ra;
        P.seed = B.seed;
        P.props = B.props;
        P.ltab = B.ltab;
    }
    // (measured and dropped: the window centre from the block index -- the grid as the launch's
    // lattice, no work-list entry to wait for: 10.87 against 10.89 us per launch at 64^3, 12.22
    // against 12.27 for an 8x1 strip; the scalar load of the entry hides behind the table loads)
    const int4 ent = P.spx[blk];
    const int y = ent.x, x = ent.y;  // may lie outside the cube when virtual
    const bool real = ent.z != 0;
    const int sp = y * P.W + x;
    const int n_lay = min(P.n_lay, M);  // (the launcher checks P.n_lay <= M)
    const long slot = (long)(y / P.fh) * P.slots_x + x / P.fw;
    // a virtual position only matters to a launch that writes the residual back
    if (!real && !(n_lay > 0 && P.write_back)) {
        if (tid < Dp && y >= 0 && y < P.H && x >= 0 && x < P.W) P.Gcur[slot * Dp + tid] = 0.0;
        return;
    }
    // per pending layer: the <= 2 x 2 spaxels of its colour class that cover this window (raw
    // lattice coordinates: rows up to sy_lo + fhh belong to sy_lo, the others to sy_lo + fh)
    int sy_lo[M], sx_lo[M];
    unsigned vmask[M];  // bit (2 hy + hx): that covering spaxel lies inside the cube
#pragma unroll
    for (int j = 0; j < M; ++j) {
        sy_lo[j] = sx_lo[j] = 0;
        vmask[j] = 0;
        if (j < n_lay) {
            sy_lo[j] = mh_raw_cover(y - fhh, P.lay_cy[j], P.fh, fhh);
            sx_lo[j] = mh_raw_cover(x - fhw, P.lay_cx[j], P.fw, fhw);
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int sy = sy_lo[j] + ((qq >> 1) ? P.fh : 0), sx = sx_lo[j] + ((qq & 1) ? P.fw : 0);
                if (sy >= 0 && sy < P.H && sx >= 0 && sx < P.W) vmask[j] |= 1u << qq;
            }
        }
    }
    // window rows / columns inside the launch's domain
    const int r_lo = max(0, P.dy0 - (y - fhh)), r_hi = min(P.fh, P.dy1 - (y - fhh));
    const int c_lo = max(0, P.dx0 - (x - fhw)), c_hi = min(P.fw, P.dx1 - (x - fhw));
    const unsigned rowmask = r_hi > r_lo ? ((1u << r_hi) - 1u) & ~((1u << r_lo) - 1u) : 0u;
    const unsigned colmask = c_hi > c_lo ? ((1u << c_hi) - 1u) & ~((1u << c_lo) - 1u) : 0u;
    // ---- loads, in the order the setup needs them (a wavefront's loads return in order) ----
    double gv[M][K];
#pragma unroll
    for (int j = 0; j < M; ++j)
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int i = tid + k * NS;
            gv[j][k] = 0.0;
            if (j < n_lay && i < 4 * Dp) {
                const int qq = i / Dp, z = i - qq * Dp;
                const int sy = sy_lo[j] + ((qq >> 1) ? P.fh : 0), sx = sx_lo[j] + ((qq & 1) ? P.fw : 0);
                // (a masked spaxel there left a zero row; rows of lattice points outside the cube
                // are never applied)
                if ((vmask[j] >> qq) & 1u)
                    gv[j][k] = P.lay_G[j][((long)(sy / P.fh) * P.slots_x + sx / P.fw) * Dp + z];
            }
        }
    // this thread's channel of the update's lines, and the proposal (sweep tables)
    const int wave = tid >> 6;
    const int nwc = (Dp + 63) >> 6;  // wavefronts that hold channels in the tail
    double EO = 0.0, EN = 0.0;
    MHProposal q = {};
    U2 u_gibbs = {0.5, 0.5};
    if (real && wave < nwc) {
        if (tid < Dp) {
            EO = P.ltab[((long)sp * 2 + 0) * Dp + tid];
            EN = P.ltab[((long)sp * 2 + 1) * Dp + tid];
        }
        q = P.props[sp];
    }

    // ---- the window pass --------------------------------------------------------------------
    const double2 vu = make_double2(P.ivar_uniform, (2 * zl + 1 < P.D) ? P.ivar_uniform : 0.0);
    const unsigned col_bytes = (unsigned)Dp * 8u;  // one voxel column (spectrum) in bytes
    const unsigned zoff = (unsigned)zl * 16u;
    const char *err_b = reinterpret_cast<const char *>(P.err);
    const char *ivar_b = reinterpret_cast<const char *>(P.ivar);
    // NTV: write-through stores through a raw buffer of THIS window (aux 16 = sc1; base = its
    // first cell, 32-bit offsets within fh + 1 rows of the cube)
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    union { double2 d; v4u i; } cv;
    const unsigned rs_off = NTV ? (unsigned)max(0L, (long)(y - fhh) * P.W + (x - fhw)) * col_bytes : 0u;
    const __amdgpu_buffer_rsrc_t err_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        P.err + (NTV ? (size_t)rs_off / 8 : 0), 0,
        NTV ? (int)(unsigned)min((long)P.H * P.W * (long)col_bytes - (long)rs_off, 0x7fffffffL) : 0, 0x00020000);
    // one position: the geometry words of its table entries -> flags, byte offset, loads
    auto request = [&](double bits0, double bits1, int pos, MHRound<U> &R, int u) {
        const int rel = __double2loint(bits0);
        const unsigned pk = (unsigned)__double2hiint(bits0);
        const unsigned dy = pk & 0xffu, dx = (pk >> 8) & 0xffu, sel = (pk >> 16) & 3u;
        const unsigned inside = pos < npos ? ((rowmask >> dy) & (colmask >> dx) & 1u) : 0u;
        const unsigned has = inside & (pk >> 18) & (vmask[0] >> sel) & 1u;
        unsigned fl = inside | (has << 1) | (sel << 2) | ((unsigned)min(pos, npos - 1) << 8);
        if constexpr (M == 2) {
            const unsigned pk1 = (unsigned)__double2hiint(bits1);
            const unsigned sel1 = (pk1 >> 16) & 3u;
            const unsigned has1 = inside & (pk1 >> 18) & (vmask[1] >> sel1) & 1u;
            fl |= (has1 << 4) | (sel1 << 5);
        }
        R.fl[u] = fl;
        const unsigned vox = inside ? (unsigned)(sp + rel) : 0u;
        const unsigned off = vox * col_bytes + zoff;
        R.off[u] = off;
        R.e[u] = *reinterpret_cast<const double2 *>(err_b + off);
        R.v[u] = vu;
        if (!UV) R.v[u] = mh_load_ivar<NTV>(reinterpret_cast<const double *>(ivar_b + off));
    };
    auto issue = [&](int r, MHRound<U> &R) {  // (table entries from the LDS copy: one wait)
        double bits[M][U];
        int pos[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            pos[u] = pos_of(r, u);
#pragma unroll
            for (int j = 0; j < M; ++j)
                bits[j][u] = s_tab[((size_t)j * npos + min(pos[u], npos - 1)) * 4 + 2];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) request(bits[0][u], bits[M - 1][u], pos[u], R, u);
    };
    double2 sA = make_double2(0.0, 0.0), sB = sA, sC = sA;
    // A round's LDS reads -- each position's taps (own, pending) and its staged G rows -- are
    // issued TOGETHER, unconditionally (clamped indices), before the arithmetic: one wait per
    // batch of UB positions instead of two dependent waits per position, which a workgroup that is
    // alone on its compute unit (one wavefront per SIMD) cannot hide.
    constexpr int UB = (NS > 512) ? (U + 1) / 2 : U;  // (the wide form's 168 registers: half rounds)
    auto consume = [&](MHRound<U> &R) {
#pragma unroll
        for (int b = 0; b < U; b += UB) {
            double2 ff[UB], gz[M][UB];
            double fp1[UB];
#pragma unroll
            for (int k = 0; k < UB; ++k) {
                const int u = b + k;
                if (u < U) {
                    const unsigned pos = R.fl[u] >> 8;
                    ff[k] = *reinterpret_cast<const double2 *>(s_tab + 4 * (size_t)pos);
                    gz[0][k] = *reinterpret_cast<const double2 *>(
                        reinterpret_cast<const char *>(s_gp) + ((R.fl[u] >> 2) & 3u) * col_bytes + zoff);
                    if constexpr (M == 2) {
                        fp1[k] = s_tab[((size_t)npos + pos) * 4 + 1];
                        gz[1][k] = *reinterpret_cast<const double2 *>(
                            reinterpret_cast<const char *>(s_gp) + (4u + ((R.fl[u] >> 5) & 3u)) * col_bytes + zoff);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < UB; ++k) {
                const int u = b + k;
                if (u >= U || !(R.fl[u] & 1u)) continue;
                double2 e = R.e[u];
                bool touched = false;
                // the pending layers, oldest first: e <- e + f G of each
                if (R.fl[u] & 2u) {
                    e.x = fma(ff[k].y, gz[0][k].x, e.x);
                    e.y = fma(ff[k].y, gz[0][k].y, e.y);
                    touched = true;
                }
                if constexpr (M == 2) {
                    if (R.fl[u] & 16u) {
                        e.x = fma(fp1[k], gz[1][k].x, e.x);
                        e.y = fma(fp1[k], gz[1][k].y, e.y);
                        touched = true;
                    }
                }
                if (touched && P.write_back) {
                    if constexpr (NTV) {
                        cv.d = e;
                        __builtin_amdgcn_raw_buffer_store_b128(cv.i, err_rsrc, (int)(R.off[u] - rs_off), 0, 16);
                    } else {
                        *reinterpret_cast<double2 *>(reinterpret_cast<char *>(P.err) + R.off[u]) = e;
                    }
                }
                D3D_ACCUM(e, R.v[u], ff[k].x);
            }
        }
    };
    MHRound<U> A;
    if (active) {  // round 0 flies during the setup (its table entries straight from memory)
#pragma unroll
        for (int u = 0; u < U; ++u) request(pe0[0][u], pe0[M - 1][u], pos_of(0, u), A, u);
    }

    // ---- setup: position tables and staged G rows into LDS ------------------------------------
#pragma unroll
    for (int j = 0; j < M; ++j) {
        if (tid < npos) *reinterpret_cast<double4 *>(s_tab + ((size_t)j * npos + tid) * 4) = t0[j];
        for (int p = tid + NS; p < npos; p += NS)  // (FSFs of more than NS taps)
            *reinterpret_cast<double4 *>(s_tab + ((size_t)j * npos + p) * 4) = T[j][p];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int i = tid + k * NS;
            if (i < 4 * Dp) s_gp[(size_t)j * 4 * Dp + i] = gv[j][k];
        }
    }
    if (real && wave < nwc && !P.ext_lines)
        u_gibbs = philox_pair(P.seed, (uint32_t)((y + P.gy0) * P.Wg + (x + P.gx0)), sweep, BLK_GIBBS);
    __syncthreads();
    D3D_MH_STAMP(blockIdx.x, 1, 0);

    if (active) {
        // (measured and dropped: a ring -- slot u requested again right after it is consumed --
        // 64^3 10.3 -> 13.1 us per launch: the table read and the address of every request then
        // sit between two consumes; and two buffers of four positions: no better than one of eight)
        for (int r = 0; r < rounds; ++r) {
            if (r > 0) issue(r, A);
            consume(A);
        }
        if (real) {
            double *rr = s_red + (size_t)g * 3 * Dp + 2 * zl;
            rr[0] = sA.x;
            rr[1] = sA.y;
            rr[Dp] = sB.x;
            rr[Dp + 1] = sB.y;
            rr[2 * Dp] = sC.x;
            rr[2 * Dp + 1] = sC.y;
        }
    }
    D3D_MH_STAMP(blockIdx.x, 2, 0);
    if (!real) {  // a masked spaxel inside the cube leaves a zero row (see k_mh_ws)
        if (tid < Dp && y >= 0 && y < P.H && x >= 0 && x < P.W) P.Gcur[slot * Dp + tid] = 0.0;
        return;
    }
    __syncthreads();  // group partial sums are in s_red
    // ---- the tail: the wavefronts that hold channels (thread t <-> channel t) -----------------
    if (wave >= nwc) return;  // (a wavefront that has ended no longer counts at a barrier)
    double sums[7], tot[7];
    mh_channel_sums_regs(P, s_red, q, tid, G, EO, EN, sums);
    bool accept;
    double r;
    if constexpr (FULL) {
        // ONE wavefront decides; the verdict travels through LDS (as k_mh_ws)
        if ((tid & 63) == 63) {
#pragma unroll
            for (int k = 0; k < 7; ++k) s_sum[wave * 8 + k] = sums[k];
        }
        __syncthreads();
        double *verdict = s_sum + 8 * nwc;
        if (wave == 0) {
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                double t = 0.0;
                for (int wv = 0; wv < nwc; ++wv) t += s_sum[wv * 8 + k];
                tot[k] = t;
            }
            D3D_MH_STAMP(blockIdx.x, 6, 0);
            mh_decide_core(P, q, sp, sweep, tot, u_gibbs, tid == 0, &accept, &r);
            if (tid == 0) {
                verdict[0] = accept ? 1.0 : 0.0;
                verdict[1] = r;
            }
        }
        __syncthreads();
        D3D_MH_STAMP(blockIdx.x, 7, 0);
        accept = verdict[0] != 0.0;
        r = verdict[1];
    } else {
        if (nwc == 1) {
#pragma unroll
            for (int k = 0; k < 7; ++k) tot[k] = 0.0 + __shfl(sums[k], 63);
        } else {
            if ((tid & 63) == 63) {
#pragma unroll
                for (int k = 0; k < 7; ++k) s_sum[wave * 8 + k] = sums[k];
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 7; ++k) {
                double t = 0.0;
                for (int wv = 0; wv < nwc; ++wv) t += s_sum[wv * 8 + k];
                tot[k] = t;
            }
        }
        D3D_MH_STAMP(blockIdx.x, 6, 0);
        mh_decide_core(P, q, sp, sweep, tot, u_gibbs, tid == 0, &accept, &r);
        D3D_MH_STAMP(blockIdx.x, 7, 0);
    }
    if (tid < Dp)
        P.Gcur[slot * Dp + tid] = (tid < P.D) ? residual_coeff(q.a_old, EO, r, accept ? EN : EO) : 0.0;
    D3D_MH_STAMP(blockIdx.x, 4, 0);
}

// The tables of one sweep: for every owned, unmasked spaxel its proposal (as k_mh_proposals)
// and the LSF-convolved unit lines of its current and proposed (c, w) -- one wavefront per
// spaxel running the code of k_mh_ws's prepare wavefront (zero-extended unit lines in a
// wave-private LDS region, mh_lsf per channel): the same bits.
// batch != NULL: the tables of R chains in one launch (grid over R x spaxels, chain-major): the
// chain's parameters, bounds, random stream and table pointers replace the arguments'.
static __global__ __launch_bounds__(256) void k_mh_line_table(MHArgs P, uint32_t sweep, int y0, int y1,
                                                               int x0, int x1, MHProposal *props,
                                                               double *ltab, const MHChainArgs *batch,
                                                               int n_chains) {
    extern __shared__ double smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int i = blockIdx.x * 4 + wave;
    const int w = x1 - x0;
    const int n_sp = (y1 - y0) * w;
    if (batch) {
        const int r = i / n_sp;
        if (r >= n_chains) return;
        i -= r * n_sp;
        const MHChainArgs &B = batch[r];
        P.params = B.params;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            P.min_b[k] = B.min_b[k];
            P.max_b[k] = B.max_b[k];
            P.amp[k] = B.amp[k];
        }
        P.seed = B.seed;
        props = const_cast<MHProposal *>(B.props);
        ltab = const_cast<double *>(B.ltab);
    }
    if (i >= n_sp) return;
    const int y = y0 + i / w, x = x0 + i % w;
    const long sp = (long)y * P.W + x;
    if (!P.mask[sp]) return;
    const int N = P.N, Dp = P.Dp;
    double *gO = smem + (size_t)wave * 2 * N, *gN = gO + N;
    const MHProposal q =
        mh_propose_from(P, P.params[sp * 3 + 0], P.params[sp * 3 + 1], P.params[sp * 3 + 2],
                        (uint32_t)((y + P.gy0) * P.Wg + (x + P.gx0)), sweep);
    if (lane == 0) props[sp] = q;
    for (int j = lane; j < N; j += 64) {
        gO[j] = (j < P.D) ? unit_gaussian((double)j, q.c_old, q.w_old) : 0.0;
        gN[j] = (j < P.D) ? unit_gaussian((double)j, q.pn[1], q.pn[2]) : 0.0;
    }
    __builtin_amdgcn_wave_barrier();  // wave-private region: LDS is in order per wave
    for (int ch = lane; ch < Dp; ch += 64) {
        double EO, EN;
        mh_lsf(P, gO, gN, ch, &EO, &EN);
        ltab[(sp * 2 + 0) * Dp + ch] = EO;
        ltab[(sp * 2 + 1) * Dp + ch] = EN;
    }
}

}  // namespace d3d
