// Internal header of libdeconv3d_hip.so: the context struct and the launch
// functions shared by the library's translation units (d3d_api.hip: C ABI and host
// logic; d3d_spatial.hip: line / LSF / FSF kernels; d3d_mh.hip: the MH-within-Gibbs
// kernels).  Not installed: the public interface is include/deconv3d_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types only: the library is dlopen()ed by d3d_comm_init

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <functional>
#include <vector>

#include "../../include/deconv3d_hip.h"
#include "d3d_kernels.h"

// The WIDE form of a small colour launch at 128 channels: eleven
// streaming wavefronts per window instead of four -- one per window row of an 11 x 11 FSF
// -- in workgroups of 768 threads.
// (round 4, same-box A/B with k_mh_small through tools/build_variant.sh -DD3D_WIDE_NS=512: eight
// wavefronts, two per SIMD, 11.95 us per launch of an 8x1 strip against 11.79 with eleven)
#ifndef D3D_WIDE_NS
#define D3D_WIDE_NS 704
#endif
constexpr int MH_WIDE_NS = D3D_WIDE_NS;
// k_mh_chain: at most this many threads per workgroup (three wavefronts per SIMD: 168
// registers each, of which a thread's column of an 11-row window takes 88)
constexpr int MH_CHAIN_NT = 768;

namespace d3dh {
// sets the thread-local message d3d_last_error() returns; returns `code`
int fail(int code, const char *fmt, ...);
}

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess)                                                           \
            return d3dh::fail(D3D_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                              __FILE__, __LINE__);                                      \
    } while (0)

#define NEED(cond, code, ...) \
    do {                      \
        if (!(cond)) return d3dh::fail(code, __VA_ARGS__); \
    } while (0)

struct d3d_ctx {
    int device = 0;
    int D = 0, H = 0, W = 0, fh = 0, fw = 0;
    int Dp = 0, HL = 0, N = 0;
    bool deep = false;  // more than 1024 channels: the z-blocked kernel forms (k_*_deep)
    long HW = 0;
    size_t cube_elems = 0;  // HW * Dp
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    double *slot[D3D_SLOT_COUNT] = {};
    double *stage = nullptr;   // D*HW doubles, host-layout staging
    double *stage2 = nullptr;  // second staging (variance)
    double *params = nullptr;  // HW*3
    double *params_alt = nullptr;  // HW*3: parameter map of d3d_simulate (not the chain state)
    uint8_t *mask = nullptr;   // HW
    double *fsf = nullptr;     // fh*fw
    int *lsf_shift = nullptr;
    double *lsf_weight = nullptr;
    int ntaps = 0;
    double *dlog = nullptr;  // HW
    double *hwbuf = nullptr; // HW scratch (chi2 map)
    double *scal = nullptr;  // small device scalars (8 doubles)
    unsigned long long *accepted = nullptr;
    int4 *spx = nullptr;  // work lists per (part, colour): real spaxels first, then virtual ones
    std::vector<int> colour_real;  // fh*fw: number of real spaxels of each colour (all parts)
    size_t spx_cap = 0;
    // A PART is a rectangle of spaxels updated together: sweep = for every phase, for
    // every part of the phase, for every colour class, one launch (lib/run.py:553-560
    // declares the scan order overridable).  An unpartitioned context has one part,
    // the whole cube (a tile: its owned rectangle).  The DOMAIN of a part is the set
    // of cells its windows touch.
    struct Part {
        int y0 = 0, y1 = 0, x0 = 0, x1 = 0;      // spaxels (local)
        int dy0 = 0, dy1 = 0, dx0 = 0, dx1 = 0;  // domain (local)
        int phase = 0;
        int layers = 1;                          // pending layers in use
        bool wide = false;                       // its colour launches take the wide form (MH_WIDE_NS)
        bool small = false;                      // its colour launches do not fill the chip: k_mh_small
        // k_mh_chain (whole sweeps of the part in one launch): slot grid, or chain = false
        bool chain = false;
        int chain_ns = 0, chain_sx0 = 0, n_sy = 0, n_sx = 0, K = 0;
        int last_col = -1;                       // its last active colour
        std::vector<int> off;                    // fh*fw + 1: start of each colour's list in spx
        std::vector<int> real;                   // fh*fw: real spaxels of each colour
    };
    std::vector<Part> parts;
    std::vector<int4> part_rects;  // as given to d3d_set_parts ({y0,y1,x0,x1}; .phase apart)
    std::vector<int> part_phase;
    int n_phases = 1;
    int pend_part = -1;            // part the pending layers belong to
    // halo plans (tiled chains): per plan a list of rectangle copies to / from peers
    struct HaloEntry {
        int peer = 0, what = 0;                   // what: 0 = SLOT_ERR cells, 1 = parameter map
        int sy0 = 0, sy1 = 0, sx0 = 0, sx1 = 0;  // rectangle sent (local), empty when sy1 <= sy0
        int ry0 = 0, ry1 = 0, rx0 = 0, rx1 = 0;  // rectangle received (local)
        size_t send_off = 0, send_n = 0, recv_off = 0, recv_n = 0;  // in doubles
    };
    std::vector<std::vector<HaloEntry>> plans;
    double *halo_send = nullptr, *halo_recv = nullptr;
    size_t halo_send_cap = 0, halo_recv_cap = 0;
    // asynchronous chain streaming (lib/run.py:447-451 chain[i] = parameters, :428-432):
    // at a saved sweep the compute stream snapshots parameters + log ratios device to
    // device (microseconds), a COPY stream moves the snapshot to a pinned host buffer,
    // and the host thread copies it into the caller's (pageable) chain while the GPU
    // runs on -- STREAM_NB snapshots may be in flight
    static constexpr int STREAM_NB = 4;
    hipStream_t copy_stream = nullptr;
    double *snap_dev[STREAM_NB] = {};   // [HW*4]: params (HW*3) | dlog (HW)
    double *snap_host[STREAM_NB] = {};  // pinned
    hipEvent_t snap_ready[STREAM_NB] = {}, snap_done[STREAM_NB] = {};
    // RCCL communicator of the tiled chain (d3d_comm_init)
    ncclComm_t comm = nullptr;
    int comm_rank = -1, comm_size = 0;
    // option halo_timing = 1: HIP events around every halo exchange of d3d_mh_sweeps
    int halo_timing = 0;
    static constexpr size_t HALO_RING = 64;  // event pairs in flight at most
    std::vector<hipEvent_t> halo_ev;   // ring of pairs (start, stop), recorded on the ctx stream
    size_t halo_ev_used = 0, halo_ev_head = 0;  // pairs in flight, oldest pair
    double halo_ms = 0.0;              // summed at the end of each d3d_mh_sweeps call
    long halo_count = 0;
    std::vector<uint8_t> h_mask;

    bool have_taps = false, have_data = false, have_params = false, have_cfg = false;
    bool err_valid = false;
    double min_b[3] = {}, max_b[3] = {}, amp[3] = {};
    double ra = 0;
    uint64_t seed = 0;
    int refresh_every = 1000;
    uint32_t sweep_origin = 0;  // Philox sweep index of sweep s is s + sweep_origin (resumed runs)

    int mh_nt = 0, mh_maxit = 0;  // MH kernel geometry (derived: apply_mh_options)
    int mh_nt_opt = 0, mh_maxit_opt = -1;  // options mh_nt / mh_maxit (0 / -1: chosen per shape)
    int mh_defer = 1;             // deferred residual write-back in effect (pick_mh_geometry: mh_defer_opt,
                                  // 0 for deep cubes that cannot take the z-blocked kernels)
    int mh_defer_opt = 1;         // option mh_defer
    int mh_zblocks = 1;           // option mh_zblocks = 0: cubes deeper than 512 channels keep k_mh_defer / k_mh_deep
    bool mh_zb = false;           // the z-blocked kernels run (k_mh_ws<..., ZBK> + k_mh_zdecide)
    double *z_part = nullptr;     // [items of a launch][blocks][4 waves][8] wave sums
    double *z_E = nullptr;        // [2][slots][Dp] lines of the updates of the last colour class
    size_t z_part_cap = 0;        // doubles allocated
#ifdef D3D_EXPERIMENTS
    unsigned long long *stampbuf = nullptr;  // D3D_MH_STAMP=1: [launch][workgroup][8]
    size_t stamp_launches = 0, stamp_next = 0, stamp_stride = 0;
#endif
    bool ivar_is_uniform = false; // SLOT_IVAR holds one constant (k_mh_ws<.., true> skips reading it)
    double ivar_uniform = 0.0;
    int uniform_fast_path = 1;    // option uniform_ivar = 0 turns the variant off
    double *gbuf[4] = {nullptr, nullptr, nullptr, nullptr};  // update coefficients [slots][Dp]
    // pending layers, oldest first: colour class and G buffer of each update that has
    // not been written into SLOT_ERR yet (k_mh_ws applies up to mh_layers of them)
    int lay_n = 0, lay_cy[3] = {-1, -1, -1}, lay_cx[3] = {-1, -1, -1}, lay_g[3] = {0, 0, 0};
    int mh_layers = 2;            // most pending layers any part uses (d3d_mh_layers)
    bool mh_nt_ivar = false;      // 1/variance loads non-temporal: residual + 1/variance exceed the Infinity Cache
    int mh_nt_ivar_opt = -1;      // option mh_nt_ivar: -1 by working set, 0 / 1 forced (within the raw buffer's 2 GiB)
    int mh_zigzag = 1;            // option mh_zigzag: odd colour ordinals walk windows / work lists backwards (MHArgs::rev)
    int mh_layers_opt = 0;        // option mh_layers: 0 chosen per part, 1..3 forced
    int mh_layers_cfg = 2;        // derived: layers of a part whose launches fill the chip
    bool mh_layers_forced = false;
    // dataflow kernel (k_mh_flow): one launch per sweep
    int mh_flow = 0;              // D3D_MH_FLOW=1: one launch per sweep (k_mh_flow; measured
                                  // slower than one k_mh_ws launch per colour: DESIGN.md)
    int flow_K = 0, flow_LY = 0, flow_LX = 0, flow_items = 0, flow_grid = 0;
    int flow_last_cy = -1, flow_last_cx = -1;  // colour class of the last active colour
    // k_mh_pair (two colour classes per launch): per-item flags with epochs and a
    // monotonic ticket counter, so that nothing needs clearing between launches
    int mh_chain_opt = 0;          // option mh_chain (EXPERIMENTS builds): 1 = k_mh_chain wherever a part's
                                   // slots are all resident; measured no faster than the colour launches
                                   // (profiles/r03_chain_phases.txt)
    int2 *chain_cols = nullptr;    // [parts][fh*fw] local residues of each part's active colours
    size_t chain_cols_cap = 0;
    unsigned *chain_flags = nullptr;  // [2][chain_slots_cap]: flag1 | flag2 (monotonic epochs)
    size_t chain_slots_cap = 0;
    double *chain_G = nullptr;     // [2][K][slots][Dp] G rows | [slots][K][2][Dp] unit lines of a sweep
    size_t chain_G_cap = 0;
    unsigned chain_base = 0;       // the epoch every flag of a finished launch holds
    bool chain_used = false;       // a chain launch ran since the error word was last read
    int mh_props = 1;              // option mh_props: the proposals of a sweep in one launch before its colours
    d3d::MHProposal *props = nullptr;  // [HW]
    long props_sweep = -1;         // the sweep (Philox number) the table holds, -1 = none
    int mh_prio = 0;               // option mh_prio: staggered completion by wave priority (MHArgs::prio)
    int mh_small = 1;              // option mh_small = 0: small colour launches keep round 3's k_mh_ws variants instead of k_mh_small
    double *ptab = nullptr;        // [(fh fw + 1)][fh fw][4] relative position tables of k_mh_small (ensure_proposals)
    bool ptab_valid = false;       // (d3d_set_taps invalidates them)
    double *ltab = nullptr;        // [HW][2][Dp] line table of the current sweep (k_mh_line_table), allocated on first use
    int mh_wide = 1;               // option mh_wide = 0: never the wide form for the small launches of a partitioned context
    int mh_pair = 0;               // D3D_MH_PAIR=1: two colour classes per launch (k_mh_pair;
                                   // measured 43.0 vs 42.0 us per colour: opt-in)
    unsigned *pair_state = nullptr;  // [0] ticket counter | [4 ..] done flags per item
    unsigned pair_epoch = 0, pair_tickets = 0;
    std::vector<int> flow_first;   // first item of every active colour (+ total)
    std::vector<int> flow_colour;  // colour class index of every active colour
    int4 *flow_ent = nullptr;     // [items] {y, x, real, colour ordinal}
    int4 *flow_col = nullptr;     // [K] {first ticket, cy, cx, -}
    int *flow_lat = nullptr;      // [K][LY*LX]
    unsigned *flow_state = nullptr;  // one block, zeroed per launch: ctl[4] | cnt[K] | done[items]
    unsigned *flow_err = nullptr;    // sticky error word of k_mh_flow
    size_t flow_state_bytes = 0, flow_cap_items = 0, flow_cap_K = 0;
    int slots_x = 0, slots = 0;
    int gy0 = 0, gx0 = 0, Wg = 0;    // tile origin / global width (RNG keys)
    bool tiled = false;              // d3d_set_tile was called
    int oy0 = 0, oy1 = 0, ox0 = 0, ox1 = 0;  // owned local rectangle
    double *prev = nullptr;          // [HW*3] parameters before each spaxel's last update
    double *recbuf = nullptr;        // [HW*8] staging of update records
    int *idxbuf = nullptr;           // [HW] staging of spaxel lists
    double *extbuf = nullptr;        // external-lines staging: [cap][6 + 2D] doubles
    size_t ext_cap = 0;              // spaxels per d3d_mh_colour_lines call it can hold
    bool fsf_sep = false;         // fsf == u v^T to rounding (k_spatial_sep); D3D_SPATIAL_SEP=0 disables
    int sep_fuse = 1;             // LSF in the same pass (k_spatial_sep_lsf); option sep_fuse = 0 disables
    int spatial_sep = 1;          // option spatial_sep = 0: treat an outer-product FSF as a general one
    double *sep_uv = nullptr;     // [fh + fw] on the device
    bool fsf_symx = false;        // fsf[k][i] == fsf[k][fw-1-i] bit for bit
    bool fsf_symy = false;        // fsf[k][i] == fsf[fh-1-k][i] bit for bit
    double *fsf_quad = nullptr;   // [(fhh+1)^2] quadrant taps of an x/y-symmetric square FSF (k_conv_rows)
    double *fsf_quad_sep = nullptr;  // the same table for an outer-product FSF: row 0 = v, row 1 = u
    bool fsf_symt = false;        // ... and fsf[k][i] == fsf[i][k] bit for bit (radial FSFs)
    bool lsf_dense_sym = false;   // dense LSF weights mirror-symmetric bit for bit
    int conv_rows = 1;            // D3D_CONV_ROWS=0: never use the one-pass kernel k_conv_rows
    int conv_zb = 1;              // option conv_zb = 0: depths above 128 keep the march kernels (no z-blocks)
    double *lsf_dense = nullptr;  // [2*LSF_RL+1] dense LSF weights for the fused epilogue
    bool lsf_dense_ok = false;    // taps within +-LSF_RL and power-of-two depth (z-major spectral kernel)
    bool lsf_dense_any = false;   // taps within +-LSF_RL channels at ANY depth (k_spectral_blocks)
    int spectral_blocks = 1;      // option spectral_blocks = 0: never use k_spectral_blocks
    int lines_dense = 1;          // option lines_dense: 0 the line cube by the tap-list kernel (k_lines); 1 by
                                  // k_lines_dense where the LSF is applied with it (k_lines' bits); 2 always,
                                  // with its own exp (within 2 ulp; measured 7-10 % faster: not the default);
                                  // 3 always, with the library's exp (tests)
    int lines_rounds = 0;         // option lines_rounds: spaxel rounds per wavefront of k_lines_dense (0: by size)
    bool lsf_fusable = false;     // taps within +-LSF_RL, power-of-two depth, strip within a wave
    int spectral_dense = 1;       // D3D_SPECTRAL_DENSE=0: always the general tap-list kernel
    int spectral_shfl = 0;        // D3D_SPECTRAL_SHFL=1: wavefront shuffles instead of the LDS window
    int fuse_lsf = 0;             // D3D_FUSE_LSF=1: LSF in the march epilogue (correct; slower today: register spills)
    int march_hy = 16;            // output rows per strip of the march kernel (derived in d3d_set_taps)
    int march_hy_opt = 0;         // option march_hy (0: chosen per shape)
    int conv_hy_opt = 0;          // option conv_hy: rows per strip of k_conv_rows (0: one strip per CU)
    int zmajor = 1;               // option zmajor = 0: d3d_convolve never uses the reference-layout kernels
    int zmajor_hy = 0;            // output rows per strip of the z-major spatial kernel (0: by shape)
    int sp_nt_opt = 0;            // option spatial_nt: workgroup size of the spatial kernels (0: by depth)
    int xcd_remap = 1, alt_dir = 1, stagger = 0;  // march kernels: workgroup -> XCD mapping, strip direction
    // host copies of the taps, so that an option that changes their analysis can redo it
    std::vector<double> h_fsf, h_lsf;
    bool h_has_lsf = false;
    double h_thr = 0.0;
    int march_one = 0;            // D3D_MARCH_ONE=2|3: one-channel-per-lane variant, TX columns
    int march_pf = 0;             // D3D_MARCH_PF=2|3: software-pipelined variant, TX columns
    // 0: tile kernel; 1: march; 2: march + the mirror symmetries the FSF has (x, and y on
    // top of x); 3: march + x symmetry only
    int march_mode = 2;
    int sp_nt = 256;              // spectral / spatial block size
    int march_stamp = 0;          // option march_stamp (EXPERIMENTS builds): phase stamps of the march kernel
};

namespace d3dh {

// ---- d3d_spatial.hip: line build, LSF and FSF passes ---------------------------------
int pick_nt(int HL);  // block size of the group-per-spaxel kernels: at least HL threads
// params: (H,W,3) map on the device (NULL: the chain state c->params)
int launch_lines(d3d_ctx *c, double *out, int convolved, const double *params = nullptr);
int launch_spectral(d3d_ctx *c, const double *in, double *out);
bool conv_rows_usable(const d3d_ctx *c, bool with_lsf);
bool can_fuse_lsf(const d3d_ctx *c);
// out = FSF (*) in, or data - FSF (*) in when data != NULL.  in != out.
// fuse_lsf: also apply the LSF along z (only when can_fuse_lsf()).
int launch_spatial(d3d_ctx *c, const double *in, double *out, const double *data,
                   bool fuse_lsf = false);
// params -> SLOT_TMP0 (LSF lines) -> dst (sim, or residual when resid)
int forward_into(d3d_ctx *c, double *dst, bool resid);
bool zmajor_ok(const d3d_ctx *c);
// LSF (x) FSF of c->stage in the reference layout (D,H,W), in place (zmajor_ok())
int launch_zmajor_convolve(d3d_ctx *c);

// ---- d3d_mh.hip: the MH-within-Gibbs kernels ---------------------------------------------
void pend_clear(d3d_ctx *c);
int pend_free_buf(const d3d_ctx *c);
void pend_push(d3d_ctx *c, int cy, int cx, int g);
void fill_mh_args(d3d_ctx *c, d3d::MHArgs &P);
int launch_mh(d3d_ctx *c, const d3d::MHArgs &P, unsigned grid, uint32_t sweep);
int launch_mh_zb(d3d_ctx *c, d3d::MHArgs &P, unsigned n_items, uint32_t sweep, int layers);
int mh_sweeps_batch(d3d_ctx **cs, int R, int n_sweeps, int first_sweep, int64_t *accepted,
                    const std::function<int(int)> &after_sweep, const std::function<int()> &drain);
int launch_mh_defer(d3d_ctx *c, const d3d::MHArgs &P, unsigned grid, uint32_t sweep, int layers,
                    bool wide);
int flush_pending(d3d_ctx *c);
// the proposals of sweep `sweep` for every owned spaxel (MHArgs::props), once per sweep
int ensure_proposals(d3d_ctx *c, uint32_t sweep);
// k_mh_small can take this context's small parts (depth, tap count, options)
bool mh_small_usable(const d3d_ctx *c);
// row of the position tables for a launch of local colour residues (ly, lx) over the pending layer
int mh_ptab_row(const d3d_ctx *c, int ly, int lx, int layer = 0);
// the part's colour launches run k_mh_small (its small form, or -- option mh_small = 2 -- the chip-filling one)
bool mh_part_uses_tables(const d3d_ctx *c, const d3d_ctx::Part &pt);
// n_sweeps whole sweeps (Philox numbers sweep0 ..) of part pi in one launch (Part::chain)
#ifdef D3D_EXPERIMENTS
int launch_mh_chain(d3d_ctx *c, int pi, uint32_t sweep0, int n_sweeps);
#endif
int launch_apply_updates(d3d_ctx *c, const d3d::MHArgs &P, const double *rec, int n);
int launch_rtnorm(d3d_ctx *c, long n, double lo, double hi, double mu, double sigma, uint64_t seed,
                  int wave_mode, double *buf);
#ifdef D3D_EXPERIMENTS
int launch_mh_flow(d3d_ctx *c, uint32_t sweep);
int launch_mh_pair(d3d_ctx *c, int ka, uint32_t sweep);
#endif

}  // namespace d3dh
