// Launchers of the MH-within-Gibbs kernels (d3d_kernels.h).  gfx950 only.
#include "d3d_ctx.h"
#include "d3d_mh_small.h"

namespace d3dh {

using namespace d3d;

void pend_clear(d3d_ctx *c) {
    c->lay_n = 0;
    c->pend_part = -1;
}

// a G buffer that holds no pending layer
int pend_free_buf(const d3d_ctx *c) {
    for (int b = 0; b < 4; ++b) {
        bool used = false;
        for (int j = 0; j < c->lay_n; ++j) used = used || c->lay_g[j] == b;
        if (!used) return b;
    }
    return 0;  // unreachable: at most 3 layers
}

void pend_push(d3d_ctx *c, int cy, int cx, int g) {
    c->lay_cy[c->lay_n] = cy;
    c->lay_cx[c->lay_n] = cx;
    c->lay_g[c->lay_n] = g;
    ++c->lay_n;
}

void fill_mh_args(d3d_ctx *c, d3d::MHArgs &P) {
    P.D = c->D;
    P.Dp = c->Dp;
    P.HL = c->HL;
    P.H = c->H;
    P.W = c->W;
    P.fh = c->fh;
    P.fw = c->fw;
    P.N = c->N;
    P.ntaps = c->ntaps;
    P.npos = c->fh * c->fw;
    P.err = c->slot[D3D_SLOT_ERR];
    P.ivar = c->slot[D3D_SLOT_IVAR];
    P.ivar_uniform = c->ivar_uniform;
    P.params = c->params;
    P.prev = c->prev;
    P.fsf = c->fsf;
    P.shift = c->lsf_shift;
    P.weight = c->lsf_weight;
    P.dlog = c->dlog;
    P.accepted = c->accepted;
    P.spx = c->spx;
    P.rev = 0;
    P.prio = c->mh_prio;
    P.props = nullptr;  // (run_part sets it for the parts whose colour launches are small)
    P.ltab = nullptr;   // (and the line table for those that run k_mh_small)
    P.fw_inv = (65536 + c->fw - 1) / c->fw;
    P.ptab = nullptr;
    P.ptab_row[0] = P.ptab_row[1] = c->fh * c->fw;
    P.batch = nullptr;  // (mh_sweeps_batch)
    P.b_items = 0;
    P.b_gcur = 0;
    P.b_lay_g[0] = P.b_lay_g[1] = P.b_lay_g[2] = 0;
    P.z_part = c->z_part;
    P.z_E = c->z_E;
    P.z_db = 256;
    P.z_nb = (c->Dp + P.z_db - 1) / P.z_db;
    P.z_slots = (int)c->slots;
    for (int k = 0; k < 3; ++k) {
        P.min_b[k] = c->min_b[k];
        P.max_b[k] = c->max_b[k];
        P.amp[k] = c->amp[k];
    }
    P.ra = c->ra;
    P.seed = c->seed;
    P.gy0 = c->gy0;
    P.gx0 = c->gx0;
    P.Wg = c->Wg;
    // the domain of the part whose layers are pending (the whole cube when none are)
    if (c->pend_part >= 0 && c->pend_part < (int)c->parts.size()) {
        const d3d_ctx::Part &pt = c->parts[c->pend_part];
        P.dy0 = pt.dy0;
        P.dy1 = pt.dy1;
        P.dx0 = pt.dx0;
        P.dx1 = pt.dx1;
    } else {
        P.dy0 = 0;
        P.dy1 = c->H;
        P.dx0 = 0;
        P.dx1 = c->W;
    }
    P.mask = c->mask;
    P.n_lay = c->lay_n;
    P.write_back = 1;
    for (int j = 0; j < 3; ++j) {
        const bool live = j < c->lay_n;
        P.lay_cy[j] = live ? c->lay_cy[j] : -1;
        P.lay_cx[j] = live ? c->lay_cx[j] : -1;
        P.lay_G[j] = c->gbuf[live ? c->lay_g[j] : 0];
    }
    P.Gcur = c->gbuf[pend_free_buf(c)];
    // the kernels that keep one pending layer (k_mh_defer, k_mh_flow) see the newest
    const int last = c->lay_n - 1;
    P.Gprev = c->gbuf[last >= 0 ? c->lay_g[last] : 0];
    P.prev_cy = last >= 0 ? c->lay_cy[last] : -1;
    P.prev_cx = last >= 0 ? c->lay_cx[last] : -1;
    P.slots_x = c->slots_x;
    P.ext_idx = nullptr;
    P.ext_in = nullptr;
    P.ext_lines = nullptr;
    P.ext_out = nullptr;
    P.ext_gibbs = 1;
    P.probe = 0;
    P.probe_sp = 0;
    P.probe_p[0] = P.probe_p[1] = P.probe_p[2] = 0.0;
    P.probe_out = c->scal;
#ifdef D3D_EXPERIMENTS
    P.stamp = nullptr;
#endif
}

template <int NT, int MAXIT>
int launch_mh_t(d3d_ctx *c, const d3d::MHArgs &P, unsigned grid, uint32_t sweep) {
    const size_t lds = d3d::mh_lds_doubles(NT, c->HL, c->Dp, c->N, P.npos) * sizeof(double);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(d3d::k_mh<NT, MAXIT>), dim3(grid), dim3(NT), lds, c->stream,
                       P, sweep);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int NT>
int launch_mh_nt(d3d_ctx *c, const d3d::MHArgs &P, unsigned grid, uint32_t sweep) {
#ifdef D3D_EXPERIMENTS
    // register-resident windows (option mh_maxit): measured slower -- occupancy
    switch (c->mh_maxit) {
        case 4: return launch_mh_t<NT, 4>(c, P, grid, sweep);
        case 8: return launch_mh_t<NT, 8>(c, P, grid, sweep);
        case 16: return launch_mh_t<NT, 16>(c, P, grid, sweep);
        case 32: return launch_mh_t<NT, 32>(c, P, grid, sweep);
        default: break;
    }
#endif
    return launch_mh_t<NT, 0>(c, P, grid, sweep);
}

int launch_mh(d3d_ctx *c, const d3d::MHArgs &P, unsigned grid, uint32_t sweep) {
    if (c->deep) {  // more than 1024 channels: threads loop over their z-pairs
        const size_t lds = d3d::mh_deep_lds_doubles(c->N, P.npos) * sizeof(double);
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(d3d::k_mh_deep),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(d3d::k_mh_deep, dim3(grid), dim3(1024), lds, c->stream, P, sweep);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    switch (c->mh_nt) {
        case 128: return launch_mh_nt<128>(c, P, grid, sweep);
        case 256: return launch_mh_nt<256>(c, P, grid, sweep);
        case 512: return launch_mh_nt<512>(c, P, grid, sweep);
        default: return launch_mh_nt<1024>(c, P, grid, sweep);
    }
}

template <int NT>
int launch_mh_defer_nt(d3d_ctx *c, const d3d::MHArgs &P, unsigned grid, uint32_t sweep) {
    const size_t lds = d3d::mh_lds_doubles(NT, c->HL, c->Dp, c->N, P.npos) * sizeof(double);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(d3d::k_mh_defer<NT>), dim3(grid), dim3(NT), lds, c->stream,
                       P, sweep);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <bool UV, int U, int M, int K, bool NTV = false, int NS = 256>
int launch_mh_ws_um(d3d_ctx *c, const d3d::MHArgs &P, unsigned grid, uint32_t sweep) {
    // (round 4, measured: the workgroups of a small launch already run on distinct compute
    // units -- HW_ID stamps, tools/mh_tail.py -- so asking for more than half of a CU's LDS to
    // force that changes nothing)
    const size_t lds = d3d::mh_ws_lds_doubles(NS, c->HL, c->Dp, c->N, P.npos, M) * sizeof(double);
    hipError_t attr = hipSuccess;
    auto go = [&](auto kern) {
        // (more than 64 KB of dynamic LDS -- 512 channels with two pending layers -- has to be
        // allowed per function)
        if (lds > 65536)
            attr = hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NS + 64), lds, c->stream, P, sweep);
    };
    // (the number of pending layers as a template constant: see k_mh_ws)
    switch (P.n_lay <= M ? P.n_lay : -1) {
        case 0: go(d3d::k_mh_ws<NS, UV, U, M, K, 0, NTV>); break;
        case 1: go(d3d::k_mh_ws<NS, UV, U, M, K, 1, NTV>); break;
        case 2:
            if constexpr (M >= 2) go(d3d::k_mh_ws<NS, UV, U, M, K, 2, NTV>);
            break;
        case 3:
            if constexpr (M >= 3) go(d3d::k_mh_ws<NS, UV, U, M, K, 3, NTV>);
            break;
        default:
            return fail(D3D_ERR_STATE, "internal: %d pending layers for a %d-layer kernel", P.n_lay, M);
    }
    HIP_TRY(attr);
    HIP_TRY(hipGetLastError());
    return 0;
}

// k_mh_small (d3d_mh_small.h): colour launches that do not fill the chip, one pending layer.
// 256 streaming threads, or -- `wide`, 128 channels in a partitioned context -- 704.
bool mh_small_usable(const d3d_ctx *c) {
    // thread t <-> channel t in the tail; window rows / columns as bits of a 32-bit mask;
    // 32-bit byte offsets into the residual
    return c->mh_small && c->mh_props && c->mh_defer == 1 && !c->mh_zb && !c->deep && c->Dp <= 256 &&
           c->fh <= 31 && c->fw <= 31 && (double)c->cube_elems * 8.0 < 4294967296.0;
}

int mh_ptab_row(const d3d_ctx *c, int ly, int lx, int layer) {
    if (layer >= c->lay_n) return c->fh * c->fw;  // nothing pending there
    const int fhh = (c->fh - 1) / 2, fhw = (c->fw - 1) / 2;
    // any window of the class: the offset of the covering pending spaxel is the same for all
    const int y = ly + 4 * c->fh, x = lx + 4 * c->fw;
    const int oy = d3d::mh_raw_cover(y - fhh, c->lay_cy[layer], c->fh, fhh) - (y - fhh);
    const int ox = d3d::mh_raw_cover(x - fhw, c->lay_cx[layer], c->fw, fhw) - (x - fhw);
    return (oy + fhh) * c->fw + (ox + fhw);
}

bool mh_part_uses_tables(const d3d_ctx *c, const d3d_ctx::Part &pt) {
    if (!mh_small_usable(c)) return false;
    if (pt.small) return true;
#ifdef D3D_EXPERIMENTS
    // the chip-filling form (option mh_small = 2): one or two pending layers, 256 streaming
    // threads.  Bit-identical to k_mh_ws and measured SLOWER than it -- 43.7 against 40.6 us per
    // launch at 300x300x128, 79.6 against 76.8 at 256 channels, 27.7 against 26.5 at 64
    // (profiles/r04_table_kernel_full.txt): a launch that fills the chip is bound by HBM, not by
    // the instruction issue the tables save, and the tables themselves are 3 % more traffic.
    return c->mh_small == 2 && pt.layers <= 2 && !pt.wide;
#else
    return false;
#endif
}

// The relative position tables of k_mh_small (d3d_mh_small.h: MHPos), once per set of taps.
static int ensure_ptab(d3d_ctx *c) {
    if (c->ptab_valid) return 0;
    const int fh = c->fh, fw = c->fw, npos = fh * fw, fhh = (fh - 1) / 2, fhw = (fw - 1) / 2;
    std::vector<double> tab((size_t)(npos + 1) * npos * 4, 0.0);
    for (int row = 0; row <= npos; ++row) {
        const int oy = row / fw - fhh, ox = row % fw - fhw;
        for (int p = 0; p < npos; ++p) {
            const int dy = p / fw, dx = p % fw;
            int sel = 0, has = 0;
            double fp = 0.0;
            if (row < npos) {
                const int hy = dy > oy + fhh, hx = dx > ox + fhw;
                const int trow = dy - oy - hy * fh + fhh, tcol = dx - ox - hx * fw + fhw;
                sel = 2 * hy + hx;
                has = 1;
                fp = c->h_fsf[(size_t)trow * fw + tcol];
            }
            const int rel = (dy - fhh) * c->W + (dx - fhw);
            const unsigned pk = d3d::mh_pos_pack(dy, dx, sel, has);
            const unsigned long long bits = ((unsigned long long)pk << 32) | (unsigned)rel;
            double first;
            memcpy(&first, &bits, sizeof first);
            double *e = &tab[((size_t)row * npos + p) * 4];
            e[0] = c->h_fsf[p];
            e[1] = fp;
            e[2] = first;
        }
    }
    if (!c->ptab) HIP_TRY(hipMalloc(&c->ptab, tab.size() * sizeof(double)));
    HIP_TRY(hipMemcpyAsync(c->ptab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));  // `tab` goes out of scope
    c->ptab_valid = true;
    return 0;
}

template <bool UV, int NS, int U, int K, int M = 1, bool FULL = false, bool NTV = false>
int launch_mh_small_t(d3d_ctx *c, const d3d::MHArgs &P, unsigned grid, uint32_t sweep) {
    const size_t lds = d3d::mh_small_lds_doubles(NS, c->HL, c->Dp, P.npos, M) * sizeof(double);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(d3d::k_mh_small<NS, UV, U, K, false, M, FULL, NTV>), dim3(grid), dim3(NS),
                       lds, c->stream, P, sweep);
    HIP_TRY(hipGetLastError());
    return 0;
}

#ifdef D3D_EXPERIMENTS
// The chip-filling form (EXPERIMENTS builds, option mh_small = 2): one or two pending layers, two
// positions in flight per thread, ONE deciding wavefront; beyond the Infinity Cache the
// non-temporal / write-through policy.
template <bool UV, int K>
int launch_mh_small_full(d3d_ctx *c, const d3d::MHArgs &P, unsigned grid, uint32_t sweep, int layers) {
    bool ntv = false;
    if constexpr (!UV) ntv = c->mh_nt_ivar;
    if (layers >= 2) {
        if constexpr (!UV)
            if (ntv) return launch_mh_small_t<UV, 256, 2, K, 2, true, true>(c, P, grid, sweep);
        return launch_mh_small_t<UV, 256, 2, K, 2, true, false>(c, P, grid, sweep);
    }
    if constexpr (!UV)
        if (ntv) return launch_mh_small_t<UV, 256, 2, K, 1, true, true>(c, P, grid, sweep);
    return launch_mh_small_t<UV, 256, 2, K, 1, true, false>(c, P, grid, sweep);
}
#endif

template <bool UV>
int launch_mh_small(d3d_ctx *c, const d3d::MHArgs &P, unsigned grid, uint32_t sweep, bool wide, int layers,
                    bool full) {
    NEED(P.n_lay <= layers && layers <= 2 && P.ltab && P.props, D3D_ERR_STATE,
         "internal: k_mh_small with %d pending layers of %d", P.n_lay, layers);
#ifdef D3D_EXPERIMENTS
    if (full) return launch_mh_small_full<UV, 4>(c, P, grid, sweep, layers);
#else
    (void)full;
#endif
    NEED(layers == 1, D3D_ERR_STATE, "internal: the small form keeps one pending layer");
    // Eight window positions (sixteen loads) in flight per thread, six in the wide form (704
    // threads leave 168 registers; eleven spilled: 15.9 us per launch against 11.9).  Measured
    // beside it at 64^3: four 11.0 us, eight 10.5, sixteen (250 registers) 10.4.
    if (wide) return launch_mh_small_t<UV, MH_WIDE_NS, (MH_WIDE_NS > 512 ? 6 : 8), 1>(c, P, grid, sweep);
    if (c->Dp <= 64) return launch_mh_small_t<UV, 256, 8, 1>(c, P, grid, sweep);
    if (c->Dp <= 128) return launch_mh_small_t<UV, 256, 8, 2>(c, P, grid, sweep);
    return launch_mh_small_t<UV, 256, 8, 4>(c, P, grid, sweep);
}

// A launch that does not fill the chip (fewer workgroups than 2 per CU) is
// latency-bound: four window positions in flight per wavefront instead of one.
// The kernels for several pending layers need more LDS; with one layer
// configured the lean variant runs.  The pending G rows of a layer (4*Dp values)
// are staged in 2 registers per thread up to Dp = 160, in 4 beyond.
// layers: the pending layers the PART being updated uses (Part::layers) -- the kernel
// family follows the part, not the context: a small part and a chip-filling part of one
// context take different ones.
template <bool UV>
int launch_mh_ws(d3d_ctx *c, const d3d::MHArgs &P, unsigned grid, uint32_t sweep, int layers,
                 bool wide) {
    // (the uniform-variance variant also gains from the deeper queue at full size:
    // 35.2 -> 33.3 us per colour; the general one loses, 43.5 -> 49.7)
    const bool small = UV || grid < (unsigned)c->flow_grid / 2;
    // (round 4: the parts whose launches do not fill the chip -- run_part hands them the line table)
    if (P.ltab && layers <= 2)
        return launch_mh_small<UV>(c, P, grid, sweep, wide, layers, !(grid < (unsigned)c->flow_grid / 2));
    // 257 .. 512 channels (round 3): the same kernel with 512 streaming threads (thread <->
    // channel in the tail; the staged G rows, 4 Dp <= 4 x 576, in four registers); the position
    // groups (512 / HL) are those of k_mh_defer<512>, so the chain stays bit-identical to it
    if (c->Dp > 256) {
        bool ntv = false;
        if constexpr (!UV) ntv = c->mh_nt_ivar && !(grid < (unsigned)c->flow_grid / 2);
        const bool few = grid < (unsigned)c->flow_grid / 2;
        if (layers >= 2) {
            if (few) return launch_mh_ws_um<UV, 4, 2, 4, false, 512>(c, P, grid, sweep);
            // (four positions in flight: 300x300x320 106 us per launch against 111 with two and 141
            // with one; 300x300x512 186 / 187 / 225 -- two workgroups per CU, two rounds)
            if constexpr (!UV)
                if (ntv) return launch_mh_ws_um<UV, 4, 2, 4, true, 512>(c, P, grid, sweep);
            return launch_mh_ws_um<UV, 4, 2, 4, false, 512>(c, P, grid, sweep);
        }
        if (few) return launch_mh_ws_um<UV, 4, 1, 4, false, 512>(c, P, grid, sweep);
        if constexpr (!UV)
            if (ntv) return launch_mh_ws_um<UV, 1, 1, 4, true, 512>(c, P, grid, sweep);
        return launch_mh_ws_um<UV, 1, 1, 4, false, 512>(c, P, grid, sweep);
    }
    // (with several layers most launches only read: two positions in flight pay at
    // full size, 43.3 -> 42.6 us per colour; chosen per kind of launch instead -- four for
    // the read-only launches, or one for the storing ones -- measures 45.3 / 41.3 us
    // against 40.8 with two for both.  Round 2, for launches that do not fill the
    // chip: EIGHT positions in flight measured slower than four -- a 150x300 tile part 6.41
    // vs 5.26 ms per sweep, 32x16x16 10.45 vs 9.95 us per launch (165-175 VGPRs); more
    // streaming wavefronts per window: see the wide form below)
    // (1/variance with the non-temporal hint when the context's working set exceeds the
    // Infinity Cache: mh_load_ivar; only the chip-filling launches have the variant)
    if constexpr (!UV) {
        if (c->mh_nt_ivar && !small) {
            if (layers >= 3) return launch_mh_ws_um<UV, 2, 3, 2, true>(c, P, grid, sweep);
            if (layers == 2) {
                if (c->Dp > 160) return launch_mh_ws_um<UV, 2, 2, 4, true>(c, P, grid, sweep);
                return launch_mh_ws_um<UV, 2, 2, 2, true>(c, P, grid, sweep);
            }
            return launch_mh_ws_um<UV, 1, 1, 4, true>(c, P, grid, sweep);
        }
    }
    if (layers >= 3) {  // Dp <= 160
        if (small) return launch_mh_ws_um<UV, 4, 3, 2>(c, P, grid, sweep);
        return launch_mh_ws_um<UV, 2, 3, 2>(c, P, grid, sweep);
    }
    if (layers == 2) {
        if (c->Dp > 160) {
            if (small) return launch_mh_ws_um<UV, 4, 2, 4>(c, P, grid, sweep);
            return launch_mh_ws_um<UV, 2, 2, 4>(c, P, grid, sweep);
        }
        if (small) return launch_mh_ws_um<UV, 4, 2, 2>(c, P, grid, sweep);
        return launch_mh_ws_um<UV, 2, 2, 2>(c, P, grid, sweep);
    }
    // The small launches of a PARTITIONED context (tiles, d3d_set_parts) at 128 channels: a
    // launch of at most one workgroup per CU is bound by how fast ONE workgroup gets through
    // its window (121 positions through four wavefronts, ~1 us per round trip), so eleven
    // streaming wavefronts instead of four (k_mh_ws<MH_WIDE_NS = 704>, one per window row of an
    // 11 x 11 FSF, ONE position in flight per wavefront): an 8x1 rank of 300x300x128
    // 4.18 -> 3.48 ms per sweep (round 2's fifteen wavefronts, k_mh_ws<960>: 3.52; two / four
    // positions in flight: 3.61 / 3.77).  (What then bounds such a launch, by the phase
    // stamps: setup 2.3 us, the prepare wavefront's proposal -> line -> LSF chain 6.4, the
    // decision tail 4.5, the kernel boundary 2.4; a second prepare wavefront for the
    // current line gained 1.5 %: not kept.)  Another grouping of the window sums than the 256-thread form
    // (results agree to rounding, not bit for bit), hence only where nothing is compared bit for
    // bit with another scheme: a given part always takes the same form, so a tiled chain and
    // the single context given the same parts still agree to the last bit.  Shallow cubes lose
    // (32 channels: 9.9 -> 12.2 us per launch).  Option mh_wide = 0: off.
    // (decided per PART, Part::wide, so that every colour of a part groups the window sums alike)
    // (round 4, measured and dropped: ALL of a thread's positions requested before the setup --
    // U = 16 / 8 / 11 here -- 64^3 12.5 -> 13.2 us per launch, an 8x1 strip 14.2 -> 16.6: this
    // kernel's window pass is bound by instruction issue, not by round trips; k_mh_small,
    // d3d_mh_small.h, is what came of it)
    if (small && wide) return launch_mh_ws_um<UV, 1, 1, 1, false, MH_WIDE_NS>(c, P, grid, sweep);
    if (small) return launch_mh_ws_um<UV, 4, 1, 4>(c, P, grid, sweep);
    return launch_mh_ws_um<UV, 1, 1, 4>(c, P, grid, sweep);
}

#ifdef D3D_EXPERIMENTS
// One sweep in one launch (k_mh_flow).  P carries the pending colour of the
// previous sweep; afterwards the last active colour of this one is pending.
template <bool UV>
int launch_mh_flow_t(d3d_ctx *c, const d3d::MHArgs &P, const d3d::MHFlow &F, uint32_t sweep) {
    constexpr int NS = 256;
    size_t lds = d3d::mh_ws_lds_doubles(NS, c->HL, c->Dp, c->N, P.npos) * sizeof(double);
    lds += 16;  // the ticket
    hipLaunchKernelGGL(HIP_KERNEL_NAME(d3d::k_mh_flow<NS, UV>), dim3((unsigned)c->flow_items),
                       dim3(NS + 64), lds, c->stream, P, F, sweep);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_mh_flow(d3d_ctx *c, uint32_t sweep) {
    HIP_TRY(hipMemsetAsync(c->flow_state, 0, c->flow_state_bytes, c->stream));
    d3d::MHArgs P;
    fill_mh_args(c, P);
    P.rev = c->mh_zigzag;  // zig-zag enabled: the kernel derives each item's direction
    d3d::MHFlow F;
    F.ent = c->flow_ent;
    F.col = c->flow_col;
    F.lat = c->flow_lat;
    F.ctl = c->flow_state;
    F.cnt = c->flow_state + 4;
    F.done = c->flow_state + 4 + c->flow_cap_K;
    F.err = c->flow_err;
    for (int b = 0; b < 3; ++b) F.gbuf[b] = c->gbuf[b];
    F.K = c->flow_K;
    F.LY = c->flow_LY;
    F.LX = c->flow_LX;
    F.pb = c->lay_n ? c->lay_g[c->lay_n - 1] : 0;  // (one layer at most: launch_mh_flow's caller)
    F.items = c->flow_items;
    F.epoch = 1;
    const int rc = (c->ivar_is_uniform && c->uniform_fast_path)
                       ? launch_mh_flow_t<true>(c, P, F, sweep)
                       : launch_mh_flow_t<false>(c, P, F, sweep);
    if (rc) return rc;
    pend_clear(c);
    pend_push(c, c->flow_last_cy, c->flow_last_cx, (F.pb + c->flow_K) % 3);
    return 0;
}

// Colours ka (N: one layer pending, nothing written) and ka+1 (W) of the active-colour
// list in ONE launch (k_mh_pair).  P carries the one pending layer.
template <bool UV, int U, int K>
int launch_mh_pair_t(d3d_ctx *c, const d3d::MHArgs &P, const d3d::MHPair &F, uint32_t sweep) {
    constexpr int NS = 256;
    const size_t lds = d3d::mh_ws_lds_doubles(NS, c->HL, c->Dp, c->N, P.npos, 2) * sizeof(double);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(d3d::k_mh_pair<NS, UV, U, K>), dim3((unsigned)(F.n_a + F.n_b)),
                       dim3(NS + 64), lds, c->stream, P, F, sweep);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_mh_pair(d3d_ctx *c, int ka, uint32_t sweep) {
    d3d::MHArgs P;
    fill_mh_args(c, P);
    P.rev = c->mh_zigzag;  // zig-zag enabled: the kernel derives each item's direction
    d3d::MHPair F;
    F.ent = c->flow_ent;
    F.lat = c->flow_lat;
    F.ctl = c->pair_state;
    F.done = c->pair_state + 4;
    F.err = c->flow_err;
    F.first_a = c->flow_first[ka];
    F.n_a = c->flow_first[ka + 1] - c->flow_first[ka];
    F.first_b = c->flow_first[ka + 1];
    F.n_b = c->flow_first[ka + 2] - c->flow_first[ka + 1];
    F.ka = ka;
    F.a_cy = c->flow_colour[ka] / c->fw;
    F.a_cx = c->flow_colour[ka] % c->fw;
    F.LY = c->flow_LY;
    F.LX = c->flow_LX;
    F.ticket_base = c->pair_tickets;
    F.epoch = ++c->pair_epoch;
    const int ga = pend_free_buf(c);
    int gb = 0;
    for (int b = 0; b < 4; ++b) {
        bool used = b == ga;
        for (int j = 0; j < c->lay_n; ++j) used = used || c->lay_g[j] == b;
        if (!used) gb = b;
    }
    F.G_a = c->gbuf[ga];
    F.G_b = c->gbuf[gb];
    c->pair_tickets += (unsigned)(F.n_a + F.n_b);
    const bool uv = c->ivar_is_uniform && c->uniform_fast_path;
    int rc;
    if (c->Dp > 160)
        rc = uv ? launch_mh_pair_t<true, 4, 4>(c, P, F, sweep) : launch_mh_pair_t<false, 2, 4>(c, P, F, sweep);
    else
        rc = uv ? launch_mh_pair_t<true, 4, 2>(c, P, F, sweep) : launch_mh_pair_t<false, 2, 2>(c, P, F, sweep);
    if (rc) return rc;
    // afterwards colour B's updates are the only pending layer (local residues == colour
    // indices: an unpartitioned, untiled context)
    c->lay_n = 0;
    pend_push(c, c->flow_colour[ka + 1] / c->fw, c->flow_colour[ka + 1] % c->fw, gb);
    c->pend_part = 0;
    return 0;
}
#endif  // D3D_EXPERIMENTS

// Cubes deeper than MH_WS_MAX_DP channels (taps within +-8 channels): one colour class in two
// launches -- k_mh_ws<..., ZBK> on (window, 256-channel block) workgroups: the D = 256 kernel
// on every block, up to the wave sums of the decision -- and k_mh_zdecide per window: totals,
// accept, Gibbs draw, G row.  Deferred write-back and pending layers as in k_mh_ws.
template <bool UV>
int launch_mh_zb_t(d3d_ctx *c, const d3d::MHArgs &P, unsigned n_items, uint32_t sweep, int layers) {
    constexpr int NS = 256;
    const int db = P.z_db;
    const unsigned grid = n_items * (unsigned)P.z_nb;
    const bool few = grid < (unsigned)c->flow_grid / 2;
    auto go = [&](auto kern, int M) {
        // (LDS of the 256-channel kernel: 35-40 KB, four workgroups per CU)
        const size_t lds =
            d3d::mh_ws_lds_doubles(NS, db / 2, db, db + 2 * d3d::LSF_RL, P.npos, M) * sizeof(double);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NS + 64), lds, c->stream, P, sweep);
    };
    // (the variants of a 256-channel cube: two pending layers, two positions in flight, the
    // staged G rows in four registers; non-temporal 1/variance beyond the Infinity Cache)
    const bool ntv = !UV && c->mh_nt_ivar && !few;
    const int nl = P.n_lay;
    if (layers >= 2) {
        if (few) {
            if (nl == 0) go(d3d::k_mh_ws<NS, UV, 4, 2, 4, 0, false, true>, 2);
            else if (nl == 1) go(d3d::k_mh_ws<NS, UV, 4, 2, 4, 1, false, true>, 2);
            else go(d3d::k_mh_ws<NS, UV, 4, 2, 4, 2, false, true>, 2);
        } else if (ntv) {
            if constexpr (!UV) {
                if (nl == 0) go(d3d::k_mh_ws<NS, UV, 2, 2, 4, 0, true, true>, 2);
                else if (nl == 1) go(d3d::k_mh_ws<NS, UV, 2, 2, 4, 1, true, true>, 2);
                else go(d3d::k_mh_ws<NS, UV, 2, 2, 4, 2, true, true>, 2);
            }
        } else {
            if (nl == 0) go(d3d::k_mh_ws<NS, UV, 2, 2, 4, 0, false, true>, 2);
            else if (nl == 1) go(d3d::k_mh_ws<NS, UV, 2, 2, 4, 1, false, true>, 2);
            else go(d3d::k_mh_ws<NS, UV, 2, 2, 4, 2, false, true>, 2);
        }
    } else {
        if (nl == 0) go(d3d::k_mh_ws<NS, UV, 4, 1, 4, 0, false, true>, 1);
        else go(d3d::k_mh_ws<NS, UV, 4, 1, 4, 1, false, true>, 1);
    }
    HIP_TRY(hipGetLastError());
    const int nw = P.z_nb * (NS / 64);
    const size_t lds2 = ((size_t)8 * nw + 8 + 16) * sizeof(double);
    hipLaunchKernelGGL(d3d::k_mh_zdecide, dim3(n_items), dim3(256), lds2, c->stream, P, sweep, NS / 64);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_mh_zb(d3d_ctx *c, d3d::MHArgs &P, unsigned n_items, uint32_t sweep, int layers) {
    NEED(P.n_lay <= (layers >= 2 ? 2 : 1), D3D_ERR_STATE, "internal: %d pending layers for the z-blocked kernel",
         P.n_lay);
    // the blocks' wave sums of one launch and the lines of its updates.  A launch's windows
    // are distinct points of one colour class's lattice, up to one period outside the cube:
    // allocated once, for the largest launch there can be
    const size_t max_items = (size_t)(c->H / c->fh + 2) * (c->W / c->fw + 2);
    NEED((size_t)n_items <= max_items, D3D_ERR_STATE, "internal: %u windows in one colour launch", n_items);
    if (!c->z_part) {
        const size_t need = max_items * P.z_nb * 32;
        HIP_TRY(hipMalloc(&c->z_part, need * sizeof(double)));
        c->z_part_cap = need;
    }
    if (!c->z_E) HIP_TRY(hipMalloc(&c->z_E, (size_t)2 * c->slots * c->Dp * sizeof(double)));
    P.z_part = c->z_part;
    P.z_E = c->z_E;
    if (c->ivar_is_uniform && c->uniform_fast_path) return launch_mh_zb_t<true>(c, P, n_items, sweep, layers);
    return launch_mh_zb_t<false>(c, P, n_items, sweep, layers);
}

// ---- batched chains: R contexts of one geometry, ONE launch per colour class -------------
// (d3d_mh_sweeps_batch.)  The leader's arguments carry everything the chains share -- work
// list, taps, pending-layer geometry --, MHArgs::batch what differs.
template <bool UV>
int launch_mh_batch_t(d3d_ctx *c, const d3d::MHArgs &P, unsigned grid, uint32_t sweep, int layers) {
    constexpr int NS = 256;
    const bool few = grid < (unsigned)c->flow_grid / 2;
    if (P.ltab && layers == 1) {  // the joint launch does not fill the chip either: k_mh_small
        const size_t lds = d3d::mh_small_lds_doubles(NS, c->HL, c->Dp, P.npos, 1) * sizeof(double);
        auto go_small = [&](auto kern) { hipLaunchKernelGGL(kern, dim3(grid), dim3(NS), lds, c->stream, P, sweep); };
        if (c->Dp <= 64) go_small(d3d::k_mh_small<NS, UV, 8, 1, true>);
        else if (c->Dp <= 128) go_small(d3d::k_mh_small<NS, UV, 8, 2, true>);
        else go_small(d3d::k_mh_small<NS, UV, 8, 4, true>);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    auto go = [&](auto kern, int M) {
        const size_t lds = d3d::mh_ws_lds_doubles(NS, c->HL, c->Dp, c->N, P.npos, M) * sizeof(double);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(NS + 64), lds, c->stream, P, sweep);
    };
    const int nl = P.n_lay;
    if (layers >= 2) {  // Dp <= 160: the staged G rows in two registers
        if (few) {
            if (nl == 0) go(d3d::k_mh_ws<NS, UV, 4, 2, 2, 0, false, false, true>, 2);
            else if (nl == 1) go(d3d::k_mh_ws<NS, UV, 4, 2, 2, 1, false, false, true>, 2);
            else go(d3d::k_mh_ws<NS, UV, 4, 2, 2, 2, false, false, true>, 2);
        } else {
            if (nl == 0) go(d3d::k_mh_ws<NS, UV, 2, 2, 2, 0, false, false, true>, 2);
            else if (nl == 1) go(d3d::k_mh_ws<NS, UV, 2, 2, 2, 1, false, false, true>, 2);
            else go(d3d::k_mh_ws<NS, UV, 2, 2, 2, 2, false, false, true>, 2);
        }
    } else {
        if (nl == 0) go(d3d::k_mh_ws<NS, UV, 4, 1, 4, 0, false, false, true>, 1);
        else go(d3d::k_mh_ws<NS, UV, 4, 1, 4, 1, false, false, true>, 1);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int mh_sweeps_batch(d3d_ctx **cs, int R, int n_sweeps, int first_sweep, int64_t *accepted,
                    const std::function<int(int)> &after_sweep, const std::function<int()> &drain) {
    d3d_ctx *L = cs[0];  // the leader: its stream, its work lists
    const d3d_ctx::Part &pt = L->parts[0];
    const int ncol = L->fh * L->fw;
    int most = 0;
    for (int col = 0; col < ncol; ++col) most = std::max(most, pt.off[col + 1] - pt.off[col]);
    // two pending layers where the launch of all chains together fills the chip
    const int layers = (L->Dp <= 160 && (long)most * R >= L->flow_grid / 2) ? 2 : 1;
    const bool uv = L->ivar_is_uniform && L->uniform_fast_path;
    // a joint launch that still does not fill the chip: k_mh_small with the chains' sweep tables
    bool small = layers == 1;
    for (int r = 0; r < R; ++r) small = small && mh_small_usable(cs[r]);
    // every chain on the leader's stream for the duration of the call
    std::vector<hipStream_t> own(R);
    for (int r = 0; r < R; ++r) {
        HIP_TRY(hipStreamSynchronize(cs[r]->stream));
        own[r] = cs[r]->stream;
        cs[r]->stream = L->stream;
    }
    auto restore = [&]() {
        for (int r = 0; r < R; ++r) cs[r]->stream = own[r];
    };
    int rc = 0;
    d3d::MHChainArgs *dev = nullptr;
    do {
        std::vector<d3d::MHChainArgs> host(R);
        for (int r = 0; r < R && !rc; ++r) {
            d3d_ctx *c = cs[r];
            if (!c->err_valid) rc = d3d_residual(c, nullptr);
            if (!rc) rc = flush_pending(c);
            if (rc) break;
            if (hipMemsetAsync(c->accepted, 0, sizeof(unsigned long long), c->stream) != hipSuccess) rc = fail(D3D_ERR_HIP, "hipMemsetAsync");
            d3d::MHChainArgs &B = host[r];
            B.err = c->slot[D3D_SLOT_ERR];
            B.ivar = c->slot[D3D_SLOT_IVAR];
            B.ivar_uniform = c->ivar_uniform;
            B.params = c->params;
            B.prev = c->prev;
            B.dlog = c->dlog;
            B.accepted = c->accepted;
            for (int b = 0; b < 4; ++b) B.gbuf[b] = c->gbuf[b];
            for (int k = 0; k < 3; ++k) {
                B.min_b[k] = c->min_b[k];
                B.max_b[k] = c->max_b[k];
                B.amp[k] = c->amp[k];
            }
            B.ra = c->ra;
            B.seed = c->seed;
            c->props_sweep = -1;
            B.props = nullptr;
            B.ltab = nullptr;
            if (small) {
                if (!c->props && hipMalloc(&c->props, (size_t)c->HW * sizeof(d3d::MHProposal)) != hipSuccess) rc = fail(D3D_ERR_HIP, "hipMalloc");
                if (!rc && !c->ltab && hipMalloc(&c->ltab, (size_t)c->HW * 2 * c->Dp * sizeof(double)) != hipSuccess) rc = fail(D3D_ERR_HIP, "hipMalloc");
                B.props = c->props;
                B.ltab = c->ltab;
            }
        }
        if (rc) break;
        if (small) rc = ensure_ptab(L);
        if (rc) break;
        if (hipMalloc(&dev, R * sizeof(d3d::MHChainArgs)) != hipSuccess) { rc = fail(D3D_ERR_HIP, "hipMalloc"); break; }
        if (hipMemcpy(dev, host.data(), R * sizeof(d3d::MHChainArgs), hipMemcpyHostToDevice) != hipSuccess) { rc = fail(D3D_ERR_HIP, "hipMemcpy"); break; }
        for (int s = first_sweep; s < first_sweep + n_sweeps && !rc; ++s) {
            const uint32_t rs = (uint32_t)s + L->sweep_origin;
            if (small) {  // every chain's proposals and lines of this sweep, one launch
                d3d::MHArgs T;
                fill_mh_args(L, T);
                const int n = (L->oy1 - L->oy0) * (L->ox1 - L->ox0);
                const size_t lds = (size_t)4 * 2 * L->N * sizeof(double);
                hipLaunchKernelGGL(d3d::k_mh_line_table, dim3((unsigned)(((long)n * R + 3) / 4)), dim3(256), lds,
                                   L->stream, T, rs, L->oy0, L->oy1, L->ox0, L->ox1, L->props, L->ltab,
                                   (const d3d::MHChainArgs *)dev, R);
                if (hipGetLastError() != hipSuccess) { rc = fail(D3D_ERR_HIP, "k_mh_line_table"); break; }
            }
            int ord = 0;
            for (int col = 0; col < ncol && !rc; ++col) {
                if (pt.real[col] <= 0) continue;
                const int ka = ord++;
                L->pend_part = 0;
                d3d::MHArgs P;
                fill_mh_args(L, P);
                if (small) {
                    P.props = L->props;   // (replaced per chain in the kernel)
                    P.ltab = L->ltab;
                    P.ptab = L->ptab;
                    P.ptab_row[0] = mh_ptab_row(L, ((col / L->fw - L->gy0) % L->fh + L->fh) % L->fh,
                                                ((col % L->fw - L->gx0) % L->fw + L->fw) % L->fw);
                }
                P.spx = L->spx + pt.off[col];
                P.rev = (L->mh_zigzag && (ka & 1)) ? 1 : 0;
                const int n_all = pt.off[col + 1] - pt.off[col];
                P.write_back = (L->lay_n >= layers) ? 1 : 0;
                const int g_cur = pend_free_buf(L);
                P.batch = dev;
                P.b_items = n_all;
                P.b_gcur = g_cur;
                for (int j = 0; j < 3; ++j) P.b_lay_g[j] = j < L->lay_n ? L->lay_g[j] : 0;
                rc = uv ? launch_mh_batch_t<true>(L, P, (unsigned)n_all * R, rs, layers)
                        : launch_mh_batch_t<false>(L, P, (unsigned)n_all * R, rs, layers);
                if (rc) break;
                const int cy = ((col / L->fw - L->gy0) % L->fh + L->fh) % L->fh;
                const int cx = ((col % L->fw - L->gx0) % L->fw + L->fw) % L->fw;
                for (int r = 0; r < R; ++r) {  // all chains keep their pending layers alike
                    d3d_ctx *c = cs[r];
                    if (P.write_back) c->lay_n = 0;
                    pend_push(c, cy, cx, g_cur);
                    c->pend_part = 0;
                }
            }
            if (!rc && after_sweep) rc = after_sweep(s);  // (saved sweeps: snapshots of every chain)
            // lib/run.py:521-534, per chain
            for (int r = 0; r < R && !rc; ++r)
                if (cs[r]->refresh_every > 0 && s % cs[r]->refresh_every == 0)
                    rc = forward_into(cs[r], cs[r]->slot[D3D_SLOT_ERR], true);
        }
        if (rc) break;
        if (drain) rc = drain();
        if (rc) break;
        std::vector<unsigned long long> acc(R, 0);
        for (int r = 0; r < R; ++r)
            if (hipMemcpyAsync(&acc[r], cs[r]->accepted, sizeof(unsigned long long), hipMemcpyDeviceToHost, L->stream) != hipSuccess) rc = fail(D3D_ERR_HIP, "hipMemcpyAsync");
        if (hipStreamSynchronize(L->stream) != hipSuccess) rc = fail(D3D_ERR_HIP, "hipStreamSynchronize");
        if (!rc && accepted)
            for (int r = 0; r < R; ++r) accepted[r] = (int64_t)acc[r];
    } while (false);
    (void)hipStreamSynchronize(L->stream);
    if (dev) (void)hipFree(dev);
    restore();
    return rc;
}

int launch_mh_defer(d3d_ctx *c, const d3d::MHArgs &P, unsigned grid, uint32_t sweep, int layers,
                    bool wide) {
    // wave-specialised kernel: 256 (or, above 256 channels, 512) streaming threads -- thread
    // <-> channel in the tail, so D <= 512 -- + one prepare wavefront
    if (c->mh_defer == 1 && c->Dp <= d3d::MH_WS_MAX_DP) {
        if (c->ivar_is_uniform && c->uniform_fast_path)
            return launch_mh_ws<true>(c, P, grid, sweep, layers, wide);
        return launch_mh_ws<false>(c, P, grid, sweep, layers, wide);
    }
    switch (c->mh_nt) {
        case 128: return launch_mh_defer_nt<128>(c, P, grid, sweep);
        case 256: return launch_mh_defer_nt<256>(c, P, grid, sweep);
        case 512: return launch_mh_defer_nt<512>(c, P, grid, sweep);
        default: return launch_mh_defer_nt<1024>(c, P, grid, sweep);
    }
}

int ensure_proposals(d3d_ctx *c, uint32_t sweep) {
    if (c->props_sweep == (long)sweep) return 0;
    if (!c->props) HIP_TRY(hipMalloc(&c->props, (size_t)c->HW * sizeof(d3d::MHProposal)));
    d3d::MHArgs P;
    fill_mh_args(c, P);
    const int n = (c->oy1 - c->oy0) * (c->ox1 - c->ox0);
    // with the lines of every update where a part runs k_mh_small (one wavefront per spaxel)
    bool lines = false;
    for (const d3d_ctx::Part &pt : c->parts) lines = lines || mh_part_uses_tables(c, pt);
    if (lines && !c->ltab) HIP_TRY(hipMalloc(&c->ltab, (size_t)c->HW * 2 * c->Dp * sizeof(double)));
    if (lines)
        if (int rc = ensure_ptab(c)) return rc;
    if (n > 0 && lines) {
        const size_t lds = (size_t)4 * 2 * c->N * sizeof(double);
        hipLaunchKernelGGL(d3d::k_mh_line_table, dim3((unsigned)((n + 3) / 4)), dim3(256), lds, c->stream, P,
                           sweep, c->oy0, c->oy1, c->ox0, c->ox1, c->props, c->ltab,
                           (const d3d::MHChainArgs *)nullptr, 0);
        HIP_TRY(hipGetLastError());
    } else if (n > 0) {
        hipLaunchKernelGGL(d3d::k_mh_proposals, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                           P, sweep, c->oy0, c->oy1, c->ox0, c->ox1, c->props);
        HIP_TRY(hipGetLastError());
    }
    c->props_sweep = (long)sweep;
    return 0;
}

// Write the pending (deferred) residual updates into SLOT_ERR.
int flush_pending(d3d_ctx *c) {
    if (c->lay_n == 0) return 0;
    d3d::MHArgs P;
    fill_mh_args(c, P);
    const int NT = 256;
    const int S = NT / c->HL > 0 ? NT / c->HL : 1;
    const long cells = (long)(P.dy1 - P.dy0) * (P.dx1 - P.dx0);  // of the pending part's domain
    if (cells > 0) {
        if (c->HL <= 256) {
            const unsigned grid = (unsigned)((cells + S - 1) / S);
            hipLaunchKernelGGL(HIP_KERNEL_NAME(d3d::k_flush_pending<256>), dim3(grid), dim3(256), 0,
                               c->stream, P);
        } else {
            hipLaunchKernelGGL(HIP_KERNEL_NAME(d3d::k_flush_pending<1024>),
                               dim3((unsigned)cells, (unsigned)((c->HL + 1023) / 1024)), dim3(1024), 0,
                               c->stream, P);
        }
        HIP_TRY(hipGetLastError());
    }
    pend_clear(c);
    return 0;
}


#ifdef D3D_EXPERIMENTS
// ---- k_mh_chain: whole sweeps of a small part in one launch --------------------------------
namespace {
// Chain kernels of different contexts of one process must not share the chip: each needs
// ALL its workgroups resident (they wait for each other), and two half-resident grids would
// wait forever (until the kernels' time-out).  One event per device orders them.
hipEvent_t g_chain_done[64] = {};
bool g_chain_recorded[64] = {};

template <int FH, bool UV>
int launch_mh_chain_t(d3d_ctx *c, const d3d::MHArgs &P, const d3d::MHChain &F, int slots) {
    auto kern = d3d::k_mh_chain<FH, UV, MH_CHAIN_NT>;
    const int nw = (F.NS + 63) / 64;
    // one workgroup per CU (the hand-off forms are measured for that): ask for more than
    // half of the 160 KiB of LDS
    size_t lds = d3d::mh_chain_lds_doubles(c->fw, c->Dp, c->N, P.npos, F.K, nw) * sizeof(double);
    lds = std::max(lds, (size_t)82 * 1024);
    if (lds > (size_t)160 * 1024) return fail(D3D_ERR_STATE, "internal: chain kernel needs %zu B of LDS", lds);
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)slots), dim3(nw * 64 + 64), lds, c->stream, P, F);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int FH>
int launch_mh_chain_uv(d3d_ctx *c, const d3d::MHArgs &P, const d3d::MHChain &F, int slots) {
    if (c->ivar_is_uniform && c->uniform_fast_path) return launch_mh_chain_t<FH, true>(c, P, F, slots);
    return launch_mh_chain_t<FH, false>(c, P, F, slots);
}
}  // namespace

int launch_mh_chain(d3d_ctx *c, int pi, uint32_t sweep0, int n_sweeps) {
    d3d_ctx::Part &pt = c->parts[pi];
    if (!pt.chain || n_sweeps <= 0) return fail(D3D_ERR_STATE, "internal: part %d has no chain form", pi);
    if (c->lay_n)  // the kernel starts from a residual with nothing pending
        if (int rc = flush_pending(c)) return rc;
    const int slots = pt.n_sy * pt.n_sx;
    const int fhh = (c->fh - 1) / 2;
    // epochs are 32-bit and monotonic over launches: start over long before they wrap
    const unsigned span = (unsigned)n_sweeps * (unsigned)pt.K;
    if (c->chain_base > (1u << 30) || c->chain_base + span < c->chain_base) {
        HIP_TRY(hipMemsetAsync(c->chain_flags, 0, 2 * c->chain_slots_cap * sizeof(unsigned), c->stream));
        c->chain_base = 0;
    }
    c->pend_part = pi;  // fill_mh_args takes the domain from it
    d3d::MHArgs P;
    fill_mh_args(c, P);
    const int g_out = pend_free_buf(c);
    d3d::MHChain F;
    F.cols = c->chain_cols + (size_t)pi * c->fh * c->fw;
    F.flag1 = c->chain_flags;
    F.flag2 = c->chain_flags + c->chain_slots_cap;
    F.err = c->flow_err;
    F.G = c->chain_G;
    F.Gout = c->gbuf[g_out];
    F.lines = c->chain_G + (size_t)2 * pt.K * slots * c->Dp;
    F.dbg = 0;
#ifdef D3D_EXPERIMENTS
    if (const char *e = getenv("D3D_CHAIN_DBG")) F.dbg = atoi(e);  // timing-only switches (wrong results)
#endif
    F.K = pt.K;
    F.n_sy = pt.n_sy;
    F.n_sx = pt.n_sx;
    F.sy0 = pt.dy0 - fhh;
    F.sx0 = pt.chain_sx0;
    F.py0 = pt.y0;
    F.py1 = pt.y1;
    F.px0 = pt.x0;
    F.px1 = pt.x1;
    F.base = c->chain_base;
    F.sweep0 = sweep0;
    F.n_sweeps = n_sweeps;
    F.NS = pt.chain_ns;
#ifdef D3D_EXPERIMENTS
    if (c->stampbuf && (size_t)slots * pt.K * 8 <= c->stamp_launches * c->stamp_stride) P.stamp = c->stampbuf;
#endif
    const int dev = c->device & 63;
    if (!g_chain_done[dev]) HIP_TRY(hipEventCreateWithFlags(&g_chain_done[dev], hipEventDisableTiming));
    if (g_chain_recorded[dev]) HIP_TRY(hipStreamWaitEvent(c->stream, g_chain_done[dev], 0));
    int rc;
    switch (c->fh) {
        case 3: rc = launch_mh_chain_uv<3>(c, P, F, slots); break;
        case 5: rc = launch_mh_chain_uv<5>(c, P, F, slots); break;
        case 7: rc = launch_mh_chain_uv<7>(c, P, F, slots); break;
        case 9: rc = launch_mh_chain_uv<9>(c, P, F, slots); break;
        default: rc = launch_mh_chain_uv<11>(c, P, F, slots); break;
    }
    if (rc) return rc;
    HIP_TRY(hipEventRecord(g_chain_done[dev], c->stream));
    g_chain_recorded[dev] = true;
    c->chain_base += span;
    c->chain_used = true;
    // afterwards the part's last colour is the one pending layer (local residues)
    pend_clear(c);
    pend_push(c, ((pt.last_col / c->fw - c->gy0) % c->fh + c->fh) % c->fh,
              ((pt.last_col % c->fw - c->gx0) % c->fw + c->fw) % c->fw, g_out);
    c->pend_part = pi;
    return 0;
}

#endif  // D3D_EXPERIMENTS

int launch_apply_updates(d3d_ctx *c, const d3d::MHArgs &P, const double *rec, int n) {
    const size_t lds = (size_t)(2 * c->N + c->Dp) * sizeof(double);
    if (c->HL <= 256) {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(d3d::k_apply_updates<256>), dim3((unsigned)n), dim3(256),
                           lds, c->stream, P, rec, n);
    } else {
        hipLaunchKernelGGL(HIP_KERNEL_NAME(d3d::k_apply_updates<1024>), dim3((unsigned)n),
                           dim3(1024), lds, c->stream, P, rec, n);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_rtnorm(d3d_ctx *c, long n, double lo, double hi, double mu, double sigma, uint64_t seed,
                  int wave_mode, double *buf) {
    const long threads = wave_mode ? n * 64 : n;
    hipLaunchKernelGGL(d3d::k_rtnorm, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, c->stream,
                       n, lo, hi, mu, sigma, seed, wave_mode, buf);
    HIP_TRY(hipGetLastError());
    return 0;
}

}  // namespace d3dh
