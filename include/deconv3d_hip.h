/*
 * deconv3d_hip.h -- C ABI of libdeconv3d_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the per-iteration likelihood path of irap-omp/deconv3d.
 * The reference has NO FFI of its own (it is pure python/numpy),
 * so every entry point below cites the reference python code it replaces
 * (paths relative to the reference root).  The reference-side binding a
 * maintainer would add (a ctypes stub inside lib/run.py) is shown in
 * INTEGRATION.md; the build's own host side is deconv3d_amd/_lib.py.
 *
 * Conventions
 *   - all floating point is IEEE fp64; host arrays are C-contiguous and
 *     caller-owned, in the reference's layouts:
 *         cube   (D, H, W)   x fastest            lib/run.py:146-149
 *         params (H, W, 3)   (a, c, w)            lib/line_models.py:70-71
 *         fsf    (fh, fw)    odd sizes            lib/run.py:209-211
 *         lsf    [D]         centred as lib/spread_functions.py:251
 *         mask   (H, W)      uint8, 1 = iterate   lib/run.py:151-165
 *   - the library copies host data to HBM and never retains a host pointer
 *     after the call returns;
 *   - every function returns 0 on success and a negative d3d_status on error;
 *     d3d_last_error() returns a thread-local human readable message;
 *   - a d3d_ctx is bound to one device and one HIP stream; it is not
 *     thread-safe; use one ctx per device (one process per GPU in multi-GPU
 *     runs).  There is NO CPU fallback: without a HIP device d3d_ctx_create
 *     fails with D3D_ERR_NO_DEVICE.
 */
#ifndef DECONV3D_HIP_H
#define DECONV3D_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct d3d_ctx d3d_ctx;

typedef enum d3d_status {
    D3D_OK = 0,
    D3D_ERR_INVALID = -1,    /* bad argument / shape (python: ValueError)   */
    D3D_ERR_NO_DEVICE = -2,  /* no usable HIP device                        */
    D3D_ERR_HIP = -3,        /* a HIP runtime call failed                   */
    D3D_ERR_STATE = -4,      /* call sequence error (taps/data/params unset)*/
    D3D_ERR_UNSUPPORTED = -5 /* shape outside the kernels' limits           */
} d3d_status;

/* Device cubes owned by a ctx, addressable through the *_slot functions.   */
typedef enum d3d_slot {
    D3D_SLOT_DATA = 0,   /* observed cube (NaN -> 0 with ivar 0)             */
    D3D_SLOT_IVAR = 1,   /* 1/variance                                       */
    D3D_SLOT_ERR = 2,    /* residual data - sim carried by the MH loop       */
    D3D_SLOT_SIM = 3,    /* last forward model                               */
    D3D_SLOT_TMP0 = 4,   /* scratch (LSF-convolved lines / user cube)        */
    D3D_SLOT_TMP1 = 5,   /* scratch                                          */
    D3D_SLOT_COUNT = 6
} d3d_slot;

/* library version: major*10000 + minor*100 + patch */
int d3d_version(void);
/* sha256 (first 16 hex digits) of the HIP/C++ sources this binary was compiled
 * from (csrc/Makefile bakes it in); __graft_entry__.build() compares it with the
 * sources in the tree and rebuilds on a mismatch. */
const char *d3d_source_hash(void);
/* thread-local message of the last failing call ("" if none) */
const char *d3d_last_error(void);
/* number of visible HIP devices (0 and D3D_OK when there is none) */
int d3d_device_count(int *count);

/* ---- context ----------------------------------------------------------- */

/* Replaces the shape bookkeeping of Run.__init__, lib/run.py:145-149,219-224.
 * D,H,W: cube shape; fh,fw: FSF shape (odd).  Allocates all device cubes. */
int d3d_ctx_create(d3d_ctx **ctx, int device, int D, int H, int W, int fh, int fw);
int d3d_ctx_destroy(d3d_ctx *ctx);
/* Run on a caller-owned HIP stream (e.g. torch's current stream) instead of
 * the ctx's own; NULL restores the own stream. */
int d3d_ctx_set_stream(d3d_ctx *ctx, void *hip_stream);
/* Block until everything queued on the ctx stream has finished. */
int d3d_sync(d3d_ctx *ctx);
/* HIP-event stopwatch on the ctx stream (for bench.py's roofline leg). */
int d3d_timer_start(d3d_ctx *ctx);
int d3d_timer_stop(d3d_ctx *ctx, double *elapsed_ms);
/* Per-context options: which kernel family / geometry a ctx uses where the library
 * has more than one (the reference has no counterpart: numpy picks nothing,
 * lib/run.py:3-31 is its whole import list).  Every default is the measured best and
 * every choice gives the same chain (bit for bit unless DESIGN.md says "to rounding"),
 * so none is needed in normal use; they exist for the bit-identity tests and for A/B
 * measurements.  An option belongs to ONE ctx: two contexts of a process may differ.
 * The environment variable D3D_<KEY> (upper case) only supplies a NEW ctx's default.
 * Changing an option flushes pending residual updates and re-derives what depends
 * on it (work lists, tap analysis), so it may be called at any time between calls.
 * Keys: mh_defer 0|1|2, mh_zblocks, mh_layers 0(auto)|1|2|3,
 * mh_wide, mh_props, halo_timing, mh_zigzag, mh_nt_ivar -1(auto)|0|1, mh_nt, uniform_ivar, conv_rows,
 * conv_zb, conv_hy, spatial_sep, sep_fuse, spatial_mode, march_hy, zmajor, zmajor_hy,
 * spectral_dense, spectral_blocks, lines_dense, lines_rounds, spatial_nt, xcd_remap, alt_dir, stagger; a build with
 * `make EXPERIMENTS=1` adds mh_chain, mh_prio, mh_maxit, mh_flow, mh_pair, spectral_shfl, fuse_lsf, march_pf,
 * march_one, march_stamp.  Unknown key or value out of range: D3D_ERR_INVALID.
 * d3d_ctx_get_option also answers the read-only key "chain_parts": how many of the
 * ctx's parts run their sweeps as one launch of persistent workgroups (k_mh_chain). */
int d3d_ctx_set_option(d3d_ctx *ctx, const char *key, long value);
int d3d_ctx_get_option(d3d_ctx *ctx, const char *key, long *value);
/* 1 when the library was built with `make EXPERIMENTS=1` (the measured-but-not-faster
 * kernel variants of DESIGN.md section 3 are present), else 0. */
int d3d_has_experiments(void);

/* ---- inputs ------------------------------------------------------------ */

/* Taps produced once per run by instrument.fsf.as_image / lsf.as_vector,
 * lib/run.py:207-211.  lsf may be NULL (no spectral pass: lib/run.py:675-676,
 * 1015-1016).  LSF taps with |lsf[t]| <= lsf_rel_threshold*max|lsf| are
 * dropped (0 keeps every non-zero tap).  A NEGATIVE value is an error bound: the
 * smallest taps are dropped while their summed magnitude stays within
 * |lsf_rel_threshold| * sum|lsf| (the python host uses -1e-16: below the rounding of
 * the fp64 sum). */
int d3d_set_taps(d3d_ctx *ctx, const double *fsf, const double *lsf,
                 double lsf_rel_threshold);
/* Data / variance / mask setup of lib/run.py:137-200.  var may be NULL (then
 * var_scalar is the constant variance, lib/run.py:186-192); zero variances
 * become 1e12 (lib/run.py:180); NaN voxels get data 0 and 1/var 0 (SURVEY.md
 * appendix A); mask may be NULL (all ones). */
int d3d_set_data(d3d_ctx *ctx, const double *data, const double *var,
                 double var_scalar, const uint8_t *mask);
/* Current parameter map, lib/run.py:294-314,338. */
int d3d_set_params(d3d_ctx *ctx, const double *params);
int d3d_get_params(d3d_ctx *ctx, double *params);

/* ---- forward model ------------------------------------------------------ */

/* Run.simulate_clean, lib/run.py:597-621 (masked spaxels are zero). */
int d3d_build_clean(d3d_ctx *ctx, double *out_cube);
/* LSF (x) FSF convolution of an arbitrary cube: convolve_1d along z of every
 * spectrum (lib/convolution.py:89-120) then convolve2d(..., 'same') of every
 * channel (lib/run.py:1027-1029). */
int d3d_convolve(d3d_ctx *ctx, const double *in_cube, double *out_cube);
/* Fused forward model of the current parameters == Run.simulate_convolved
 * (lib/run.py:623-652) == the sim of _compute_error_in_one_step
 * (lib/run.py:999-1029).  out_sim may be NULL (result stays in SLOT_SIM). */
int d3d_forward(d3d_ctx *ctx, double *out_sim);
/* Run.simulate_clean(shape, parameters) (convolved = 0, lib/run.py:597-621) and
 * Run.simulate_convolved(shape, parameters) (convolved = 1, lib/run.py:623-652)
 * for an EXPLICIT (H,W,3) parameter map: the chain state (d3d_set_params, the
 * carried residual) is not touched.  Masked spaxels are zero. */
int d3d_simulate(d3d_ctx *ctx, const double *params, int convolved, double *out_cube);
/* err = data - forward(params), lib/run.py:334 and :525-534 / :999-1031.
 * Stored in SLOT_ERR; out_err may be NULL. */
int d3d_residual(d3d_ctx *ctx, double *out_err);
/* Per-spaxel 0.5*sum_z err^2/var of SLOT_ERR (the quantity of
 * lib/run.py:423 summed per spectrum) and its total.  Either may be NULL; with both NULL
 * the map stays on the device and the call returns without waiting for it. */
int d3d_chi2_map(d3d_ctx *ctx, double *out_hw, double *total);

/* The same convolution on a cube that stays on the device in the REFERENCE
 * layout (D,H,W) (lib/run.py:146-149): upload once, convolve in place any
 * number of times, download.  d3d_convolve == upload + convolve + download.
 * With mirror-symmetric FSFs and a compact LSF both passes run in that layout
 * (lanes along x), otherwise through the spectrum-contiguous slot kernels. */
int d3d_stage_upload(d3d_ctx *ctx, const double *cube);
int d3d_stage_convolve(d3d_ctx *ctx);
int d3d_stage_download(d3d_ctx *ctx, double *cube);

/* Device-resident variants (no host traffic; used by bench.py). */
int d3d_upload_slot(d3d_ctx *ctx, int slot, const double *cube);
int d3d_download_slot(d3d_ctx *ctx, int slot, double *cube);
int d3d_convolve_slots(d3d_ctx *ctx, int src_slot, int dst_slot);

/* ---- MH-within-Gibbs ---------------------------------------------------- */

/* Bounds (lib/run.py:235-245), Cauchy jump amplitudes (:251-262; the Gibbs
 * parameter's amplitude is forced to 0), a-priori variance of the amplitude
 * (:264-265), RNG seed (the reference is unseeded), and the cadence of the
 * from-scratch residual refresh (:525; 0 disables). */
int d3d_mh_config(d3d_ctx *ctx, const double min_b[3], const double max_b[3],
                  const double jump_amp[3], double gibbs_apriori_variance,
                  uint64_t seed, int refresh_every);
/* Resumed runs: sweep s draws the random numbers of sweep s + origin, so that a
 * chain continued from a checkpoint (lib/run.py:790-797 -> initial_parameters)
 * does not replay the random numbers of its first segment.  Default 0. */
int d3d_mh_set_sweep_origin(d3d_ctx *ctx, int64_t origin);
/* Parity probe: for a proposal p_new at spaxel (y,x) against the current
 * state, out = {ar_old, ar_new, ar_old-ar_new, sum ek^2/var, sum ek*ul/var}
 * (lib/run.py:400-426, 464-493).  Does not modify the state. */
int d3d_window_stats(d3d_ctx *ctx, int y, int x, const double p_new[3],
                     double out[5]);
/* n_sweeps sweeps of the inner loop lib/run.py:367-519 over every unmasked
 * spaxel, numbered first_sweep .. first_sweep+n_sweeps-1 (the reference's
 * cur_iteration).  After sweep s with s % keep_one_in == 0 the parameter map
 * is copied to chain_out[(s/keep_one_in)] ((H,W,3) each, lib/run.py:447-451)
 * and the log acceptance ratios to dlog_out[(s/keep_one_in)] ((H,W) each,
 * lib/run.py:428-432); either may be NULL.  *accepted (may be NULL) receives
 * the number of accepted MH proposals (lib/run.py:440). */
int d3d_mh_sweeps(d3d_ctx *ctx, int n_sweeps, int first_sweep, int keep_one_in,
                  double *chain_out, double *dlog_out, int64_t *accepted);
/* The same update for a line model evaluated on the HOST (a python LineModel
 * plugin with its own modelize(), lib/line_models.py:17-61): n spaxels of one
 * colour class (disjoint FSF windows; the caller guarantees it), per spaxel i
 *   spaxels[i]          local index y*W+x
 *   in3[i*3 + 0..2]     current Gibbs amplitude (1 if the model has none),
 *                       out-of-bounds flag of the proposal (lib/run.py:379-384),
 *                       log(u) of the acceptance test (lib/run.py:435)
 *   lines[(i*2+0)*D..]  current line, unit amplitude (lib/run.py:472, 481-488)
 *   lines[(i*2+1)*D..]  proposed line, unit amplitude
 * The device applies the LSF, the window statistics, accept, the Gibbs draw of
 * the amplitude when gibbs != 0 (bounds min_b[0]/max_b[0] of d3d_mh_config) and
 * the residual update; out3[i*3 + 0..2] = {accepted, new amplitude, delta}. */
int d3d_mh_colour_lines(d3d_ctx *ctx, int sweep, int n, const int *spaxels, const double *in3,
                        const double *lines, int gibbs, double *out3);
/* rtnorm(a, b, mu, sigma, size), lib/rtnorm.py:21-92: n draws of the normal
 * N(mu, sigma^2) truncated to [lo, hi], with the sampler the Gibbs step uses
 * (own algorithm, same distribution; the reference's Chopin tables are GPL and
 * are not reproduced).  Draw i uses the Philox stream (seed, i); wave_mode = 1
 * runs the wavefront-cooperative form of the MH kernel (bit-identical draws). */
int d3d_rtnorm(d3d_ctx *ctx, long n, double lo, double hi, double mu, double sigma,
               uint64_t seed, int wave_mode, double *out);
/* Last sweep's log acceptance ratios, (H,W). */
int d3d_get_dlog(d3d_ctx *ctx, double *out_hw);
/* *out = 1 when d3d_set_data found one constant variance and no NaN voxel -- the
 * reference's default when Run gets variance=None (lib/run.py:171-178 builds a
 * constant cube from median_clip) -- so that d3d_mh_sweeps streams the residual
 * only (16 instead of 24 bytes per window voxel; results are bit-identical to
 * the general kernel).  Option uniform_ivar = 0 turns the variant off. */
int d3d_variance_is_uniform(d3d_ctx *ctx, int *out);
/* *out = number of colours whose residual updates d3d_mh_sweeps keeps pending as
 * (colour, coefficient rows) layers before it writes the residual back: 2 by
 * default (the residual is stored every second colour: writing it costs about
 * twice what reading it does on MI355X), 1 for cubes whose colour launches do not
 * fill the chip or whose depth exceeds 256, 0 when updates are written at once (tiled contexts).  The chain
 * is bit-identical for every value.  Option mh_layers = 1|2|3 forces a depth; a
 * partitioned ctx reports the most layers any of its parts uses. */
int d3d_mh_layers(d3d_ctx *ctx, int *out);

/* ---- spatial tiling (one chain over several GPUs) --------- */
/* The reference has no counterpart (single process).  What makes tiling possible is
 * that an update at (y,x) touches only its FSF window (lib/run.py:404-419), and that
 * the scan order is overridable (lib/run.py:553-560).
 *
 * A TILE ctx holds a sub-region of the global cube: the spaxels it owns plus a
 * frame.  Random numbers and colour classes are keyed by GLOBAL coordinates.  The
 * owned spaxels are cut into PARTS (rectangles), each with a PHASE number: a sweep
 * runs the phases in order, and within a phase every part runs its fh*fw colour
 * launches.  Parts of one phase -- on this GPU or on others -- are chosen so that
 * their windows are disjoint (deconv3d_amd/tiling.py), so a phase runs on every
 * GPU at once with no communication, and after it each GPU sends the residual
 * cells it changed that a neighbour also holds: a bulk HALO copy per phase (2-4
 * per sweep, a few MB each) instead of an exchange per colour class.  The tiled
 * chain is bit-identical to a single ctx given the same parts (d3d_set_parts). */

/* Declare this ctx a tile: its (H,W) cube is the region starting at global
 * (gy0,gx0) of a cube Wg spaxels wide; it owns local rows [oy0,oy1) and columns
 * [ox0,ox1) (other spaxels are never updated here).  Call before d3d_set_data. */
int d3d_set_tile(d3d_ctx *ctx, int gy0, int gx0, int Wg, int oy0, int oy1, int ox0, int ox1);
/* Cut the owned spaxels into nparts rectangles rects[4*i..] = {y0,y1,x0,x1} (local,
 * disjoint, inside the owned rectangle) with phases[i] in [0,16).  nparts = 0
 * restores the single part.  Spaxels in no part are not updated. */
int d3d_set_parts(d3d_ctx *ctx, int nparts, const int *rects, const int *phases);
/* Several independent chains of ONE geometry in one launch per colour class (the ensemble of
 * BASELINE config 5 on one device; lib/run.py has one chain per Run(), its sweep loop is
 * lib/run.py:344-537).  ctxs[0..n_ctx): contexts on one device with the same shape, mask, FSF and
 * LSF (data, variance values, bounds, parameters, seed may all differ), unpartitioned, at
 * most 256 channels.  Every chain is the chain d3d_mh_sweeps would produce for its ctx alone
 * (same kernels, same random streams); a small cube's colour launch, which alone leaves the chip
 * idle, carries n_ctx times the windows for the same latency.  keep_one_in, chain_out[n_ctx],
 * dlog_out[n_ctx] (arrays of per-chain pointers, or NULL; a NULL entry skips that chain): as in
 * d3d_mh_sweeps, per chain.  accepted[n_ctx] (may be NULL): accepted proposals per chain. */
int d3d_mh_sweeps_batch(d3d_ctx **ctxs, int n_ctx, int n_sweeps, int first_sweep, int keep_one_in,
                        double **chain_out, double **dlog_out, int64_t *accepted);
/* One phase of sweep `sweep`: every colour class of every part of that phase
 * (lib/run.py:367-519 restricted to them).  For callers that exchange the halos
 * themselves (loop-back, host-staged transports); d3d_mh_sweeps does whole sweeps
 * including the exchange once d3d_comm_init was called. */
int d3d_mh_phase(d3d_ctx *ctx, int phase, int sweep);
/* Accepted proposals since the counter was last reset (d3d_mh_sweeps resets it). */
int d3d_mh_accepted(d3d_ctx *ctx, int64_t *count, int reset);
/* Write pending (deferred) residual updates into SLOT_ERR now. */
int d3d_flush(d3d_ctx *ctx);

/* Halo plan `plan` (0..15: after that phase; D3D_PLAN_PARAMS: the parameter gather
 * before a from-scratch residual, lib/run.py:521-534): n entries of 10 ints
 * {peer rank, kind (0 residual cells, 1 parameter map), send rectangle y0,y1,x0,x1,
 * receive rectangle y0,y1,x0,x1} in local coordinates (an empty rectangle = none). */
#define D3D_PLAN_PARAMS 16
int d3d_halo_plan(d3d_ctx *ctx, int plan, int n, const int *entries);
/* RCCL transport: rank 0 draws a unique id (D3D_COMM_UID_BYTES bytes), every rank
 * passes it to d3d_comm_init (ncclCommInitRank on the ctx's device).  Then
 * d3d_halo_exchange(plan) = pack the send rectangles, ncclSend/ncclRecv them inside
 * one ncclGroupStart/End on the ctx stream (point-to-point: one xGMI link per
 * neighbour, no ring), unpack -- device buffers only, no host synchronisation. */
#define D3D_COMM_UID_BYTES 128
int d3d_comm_unique_id(void *uid);
int d3d_comm_init(d3d_ctx *ctx, int nranks, int rank, const void *uid);
int d3d_comm_destroy(d3d_ctx *ctx);
int d3d_halo_exchange(d3d_ctx *ctx, int plan);
/* What RCCL itself reports for the ctx's communicator (ncclCommCount, ncclCommUserRank):
 * how many ranks actually joined, and which one this is (bench.py's tiled record). */
int d3d_comm_info(d3d_ctx *ctx, int *nranks, int *rank);
/* With option halo_timing = 1, d3d_mh_sweeps brackets every halo exchange (pack, RCCL
 * send/recv, unpack; lib/run.py has no counterpart: the reference is one process) with
 * HIP events on the ctx stream: *ms = their summed duration since the last reset,
 * *count (may be NULL) = how many exchanges that was. */
int d3d_halo_time(d3d_ctx *ctx, double *ms, long *count, int reset);
/* The same exchange in steps, for other transports: pack the send rectangles into
 * the plan's device send buffer / scatter the device receive buffer; device
 * pointers and sizes of one entry's buffers; host staging of one entry;
 * a device-to-device copy queued on ctx's stream (loop-back between contexts). */
int d3d_halo_pack(d3d_ctx *ctx, int plan);
int d3d_halo_unpack(d3d_ctx *ctx, int plan);
int d3d_halo_buffers(d3d_ctx *ctx, int plan, int entry, void **send_ptr, size_t *send_bytes,
                     void **recv_ptr, size_t *recv_bytes);
int d3d_halo_download(d3d_ctx *ctx, int plan, int entry, double *host);
int d3d_halo_upload(d3d_ctx *ctx, int plan, int entry, const double *host);
int d3d_device_copy(d3d_ctx *ctx, void *dst, const void *src, size_t bytes);

/* Number of owned unmasked spaxels of global colour (cy,cx) = ((y+gy0) mod fh,
 * (x+gx0) mod fw), colour = cy*fw + cx. */
int d3d_colour_count(d3d_ctx *ctx, int colour, int *count);
/* One colour class of sweep `sweep` over every part, residual written back
 * immediately (per-colour stepping for tests and probes). */
int d3d_mh_colour(d3d_ctx *ctx, int colour, int sweep);
/* Per-update records, the finer-grained alternative to the halo copies: {global y,
 * global x, a,c,w before, a,c,w after} of the last update of the n listed local
 * spaxels (y*W+x), out[n*8] ... */
int d3d_export_updates(d3d_ctx *ctx, int n, const int *spaxels, double *out);
/* ... and their replay on another ctx: the first two entries are LOCAL coordinates
 * of this tile (they may lie outside it); err += f*G on the window's part inside. */
int d3d_apply_updates(d3d_ctx *ctx, int n, const double *records);

#ifdef __cplusplus
}
#endif
#endif /* DECONV3D_HIP_H */
