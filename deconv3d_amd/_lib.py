# coding=utf-8
"""
ctypes binding of ``libdeconv3d_hip.so`` (C ABI: include/deconv3d_hip.h).

There is deliberately NO fallback: if the shared library is missing, or no HIP
device is visible when an :class:`Engine` is created, this raises.  The CPU
oracle under ``oracle/`` is test infrastructure and is never imported here.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# DECONV3D_HIP_LIB: another build of the same library (A/B measurements of kernel variants)
LIB_PATH = os.environ.get("DECONV3D_HIP_LIB") or os.path.join(_HERE, "csrc", "libdeconv3d_hip.so")

# every symbol include/deconv3d_hip.h declares (tests check the export list)
SYMBOLS = [
    "d3d_version", "d3d_source_hash", "d3d_last_error", "d3d_device_count",
    "d3d_ctx_create", "d3d_ctx_destroy", "d3d_ctx_set_stream", "d3d_sync",
    "d3d_ctx_set_option", "d3d_ctx_get_option", "d3d_has_experiments",
    "d3d_timer_start", "d3d_timer_stop",
    "d3d_set_taps", "d3d_set_data", "d3d_set_params", "d3d_get_params",
    "d3d_build_clean", "d3d_convolve", "d3d_forward", "d3d_simulate", "d3d_residual",
    "d3d_chi2_map", "d3d_upload_slot", "d3d_download_slot",
    "d3d_convolve_slots", "d3d_stage_upload", "d3d_stage_convolve", "d3d_stage_download",
    "d3d_mh_config", "d3d_mh_set_sweep_origin", "d3d_window_stats",
    "d3d_mh_sweeps", "d3d_mh_colour_lines", "d3d_get_dlog", "d3d_variance_is_uniform", "d3d_mh_layers",
    "d3d_colour_count", "d3d_rtnorm",
    "d3d_set_tile", "d3d_set_parts", "d3d_mh_phase", "d3d_mh_sweeps_batch", "d3d_mh_accepted", "d3d_flush",
    "d3d_halo_plan", "d3d_comm_unique_id", "d3d_comm_init", "d3d_comm_destroy", "d3d_comm_info",
    "d3d_halo_time",
    "d3d_halo_exchange", "d3d_halo_pack", "d3d_halo_unpack", "d3d_halo_buffers",
    "d3d_halo_download", "d3d_halo_upload", "d3d_device_copy",
    "d3d_mh_colour", "d3d_export_updates", "d3d_apply_updates",
]

PLAN_PARAMS = 16          # D3D_PLAN_PARAMS
COMM_UID_BYTES = 128      # D3D_COMM_UID_BYTES

SLOT_DATA, SLOT_IVAR, SLOT_ERR, SLOT_SIM, SLOT_TMP0, SLOT_TMP1 = range(6)

ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_STATE, ERR_UNSUPPORTED = -1, -2, -3, -4, -5


class HipLibraryMissing(ImportError):
    """libdeconv3d_hip.so has not been built (run __graft_entry__.build())."""


class HipError(RuntimeError):
    """A HIP runtime call failed, or no HIP device is available."""


_lib = None


def _dp(arr):
    return arr.ctypes.data_as(C.POINTER(C.c_double))


def load():
    """Load the shared library (once) and declare the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). deconv3d_amd has no CPU "
            "fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    ctx_p = C.c_void_p
    dbl_p = C.POINTER(C.c_double)
    lib.d3d_version.restype = C.c_int
    lib.d3d_last_error.restype = C.c_char_p
    lib.d3d_source_hash.restype = C.c_char_p
    lib.d3d_device_count.argtypes = [C.POINTER(C.c_int)]
    lib.d3d_ctx_create.argtypes = [C.POINTER(ctx_p)] + [C.c_int] * 6
    lib.d3d_ctx_destroy.argtypes = [ctx_p]
    lib.d3d_ctx_set_stream.argtypes = [ctx_p, C.c_void_p]
    lib.d3d_sync.argtypes = [ctx_p]
    lib.d3d_ctx_set_option.argtypes = [ctx_p, C.c_char_p, C.c_long]
    lib.d3d_ctx_get_option.argtypes = [ctx_p, C.c_char_p, C.POINTER(C.c_long)]
    lib.d3d_has_experiments.restype = C.c_int
    lib.d3d_timer_start.argtypes = [ctx_p]
    lib.d3d_timer_stop.argtypes = [ctx_p, dbl_p]
    lib.d3d_set_taps.argtypes = [ctx_p, dbl_p, dbl_p, C.c_double]
    lib.d3d_set_data.argtypes = [ctx_p, dbl_p, dbl_p, C.c_double,
                                 C.POINTER(C.c_uint8)]
    lib.d3d_set_params.argtypes = [ctx_p, dbl_p]
    lib.d3d_get_params.argtypes = [ctx_p, dbl_p]
    lib.d3d_build_clean.argtypes = [ctx_p, dbl_p]
    lib.d3d_convolve.argtypes = [ctx_p, dbl_p, dbl_p]
    lib.d3d_forward.argtypes = [ctx_p, dbl_p]
    lib.d3d_residual.argtypes = [ctx_p, dbl_p]
    lib.d3d_simulate.argtypes = [ctx_p, dbl_p, C.c_int, dbl_p]
    lib.d3d_mh_set_sweep_origin.argtypes = [ctx_p, C.c_int64]
    lib.d3d_chi2_map.argtypes = [ctx_p, dbl_p, dbl_p]
    lib.d3d_upload_slot.argtypes = [ctx_p, C.c_int, dbl_p]
    lib.d3d_download_slot.argtypes = [ctx_p, C.c_int, dbl_p]
    lib.d3d_convolve_slots.argtypes = [ctx_p, C.c_int, C.c_int]
    lib.d3d_stage_upload.argtypes = [ctx_p, dbl_p]
    lib.d3d_stage_convolve.argtypes = [ctx_p]
    lib.d3d_stage_download.argtypes = [ctx_p, dbl_p]
    lib.d3d_mh_config.argtypes = [ctx_p, dbl_p, dbl_p, dbl_p, C.c_double,
                                  C.c_uint64, C.c_int]
    lib.d3d_window_stats.argtypes = [ctx_p, C.c_int, C.c_int, dbl_p, dbl_p]
    lib.d3d_mh_sweeps.argtypes = [ctx_p, C.c_int, C.c_int, C.c_int, dbl_p,
                                  dbl_p, C.POINTER(C.c_int64)]
    lib.d3d_get_dlog.argtypes = [ctx_p, dbl_p]
    lib.d3d_variance_is_uniform.argtypes = [ctx_p, C.POINTER(C.c_int)]
    lib.d3d_mh_layers.argtypes = [ctx_p, C.POINTER(C.c_int)]
    lib.d3d_mh_colour_lines.argtypes = [ctx_p, C.c_int, C.c_int, C.POINTER(C.c_int), dbl_p,
                                        dbl_p, C.c_int, dbl_p]
    lib.d3d_colour_count.argtypes = [ctx_p, C.c_int, C.POINTER(C.c_int)]
    lib.d3d_rtnorm.argtypes = [ctx_p, C.c_long] + [C.c_double] * 4 + [C.c_uint64, C.c_int, dbl_p]
    lib.d3d_set_tile.argtypes = [ctx_p] + [C.c_int] * 7
    int_p = C.POINTER(C.c_int)
    lib.d3d_set_parts.argtypes = [ctx_p, C.c_int, int_p, int_p]
    lib.d3d_mh_phase.argtypes = [ctx_p, C.c_int, C.c_int]
    lib.d3d_mh_sweeps_batch.argtypes = [C.POINTER(ctx_p), C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    lib.d3d_mh_accepted.argtypes = [ctx_p, C.POINTER(C.c_int64), C.c_int]
    lib.d3d_flush.argtypes = [ctx_p]
    lib.d3d_halo_plan.argtypes = [ctx_p, C.c_int, C.c_int, int_p]
    lib.d3d_comm_unique_id.argtypes = [C.c_void_p]
    lib.d3d_comm_init.argtypes = [ctx_p, C.c_int, C.c_int, C.c_void_p]
    lib.d3d_comm_destroy.argtypes = [ctx_p]
    lib.d3d_comm_info.argtypes = [ctx_p, int_p, int_p]
    lib.d3d_halo_time.argtypes = [ctx_p, dbl_p, C.POINTER(C.c_long), C.c_int]
    lib.d3d_halo_exchange.argtypes = [ctx_p, C.c_int]
    lib.d3d_halo_pack.argtypes = [ctx_p, C.c_int]
    lib.d3d_halo_unpack.argtypes = [ctx_p, C.c_int]
    lib.d3d_halo_buffers.argtypes = [ctx_p, C.c_int, C.c_int, C.POINTER(C.c_void_p),
                                     C.POINTER(C.c_size_t), C.POINTER(C.c_void_p),
                                     C.POINTER(C.c_size_t)]
    lib.d3d_halo_download.argtypes = [ctx_p, C.c_int, C.c_int, dbl_p]
    lib.d3d_halo_upload.argtypes = [ctx_p, C.c_int, C.c_int, dbl_p]
    lib.d3d_device_copy.argtypes = [ctx_p, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.d3d_mh_colour.argtypes = [ctx_p, C.c_int, C.c_int]
    lib.d3d_export_updates.argtypes = [ctx_p, C.c_int, C.POINTER(C.c_int), dbl_p]
    lib.d3d_apply_updates.argtypes = [ctx_p, C.c_int, dbl_p]
    for name in SYMBOLS:
        fn = getattr(lib, name)
        if name not in ("d3d_version", "d3d_last_error", "d3d_source_hash", "d3d_has_experiments"):
            fn.restype = C.c_int
    _lib = lib
    return lib


def source_hash():
    """Hash of the sources the loaded binary was compiled from (csrc/Makefile)."""
    return load().d3d_source_hash().decode("ascii", "replace")


def has_experiments():
    """True when the library was built with `make EXPERIMENTS=1`."""
    return bool(load().d3d_has_experiments())


def comm_unique_id():
    """128 bytes identifying a new RCCL communicator (rank 0 draws it, every rank
    passes it to Engine.comm_init)."""
    buf = C.create_string_buffer(COMM_UID_BYTES)
    _check(load().d3d_comm_unique_id(C.cast(buf, C.c_void_p)))
    return buf.raw


def mh_sweeps_batch(engines, n_sweeps, first_sweep=1, keep_one_in=1, chains=None, dlogs=None):
    """d3d_mh_sweeps_batch: the chains of several Engines of one geometry, one launch per colour
    class for all of them.  chains / dlogs: per-engine host arrays as Engine.mh_sweeps takes
    them (or None; a None entry skips that engine).  Returns the accepted counts."""
    lib = load()
    n = len(engines)
    last = (first_sweep + n_sweeps - 1) // keep_one_in

    def pointers(arrs, tail):
        if arrs is None:
            return None
        if len(arrs) != n:
            raise ValueError("one array (or None) per engine")
        out = (C.c_void_p * n)()
        for i, (a, e) in enumerate(zip(arrs, engines)):
            if a is None:
                continue
            H, W = e.shape[1:]
            if (a.dtype != np.float64 or not a.flags.c_contiguous or a.shape[1:] != (H, W) + tail
                    or a.shape[0] <= last):
                raise ValueError("chain / dlog array %d: float64, C-contiguous, (>%d, %d, %d%s)"
                                 % (i, last, H, W, ", 3" if tail else ""))
            out[i] = a.ctypes.data
        return out

    arr = (C.c_void_p * n)(*[e._ctx.value for e in engines])
    acc = (C.c_int64 * n)()
    _check(lib.d3d_mh_sweeps_batch(arr, n, int(n_sweeps), int(first_sweep), int(keep_one_in),
                                   pointers(chains, (3,)), pointers(dlogs, ()), acc))
    return [int(v) for v in acc]


def device_count():
    lib = load()
    n = C.c_int(0)
    lib.d3d_device_count(C.byref(n))
    return n.value


def _check(rc):
    if rc == 0:
        return
    msg = load().d3d_last_error().decode("utf-8", "replace")
    if rc == ERR_INVALID:
        raise ValueError(msg)
    if rc == ERR_STATE:
        raise RuntimeError(msg)
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise HipError(msg)


def _c64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError("expected shape %s, got %s" % (tuple(shape), a.shape))
    return a


class Engine(object):
    """
    One device context: cube shape ``(D, H, W)``, FSF shape ``(fh, fw)``.
    Thin, typed mirror of the C ABI; arrays in the reference's layouts.
    """

    def __init__(self, shape, fsf_shape, device=0, options=None):
        """options: {key: int} passed to :meth:`set_option` (d3d_ctx_set_option) right
        after the context exists -- per-context, unlike the D3D_<KEY> environment
        defaults."""
        self._lib = load()
        self._ctx = C.c_void_p(None)
        D, H, W = [int(v) for v in shape]
        fh, fw = [int(v) for v in fsf_shape]
        self.shape = (D, H, W)
        self.fsf_shape = (fh, fw)
        _check(self._lib.d3d_ctx_create(C.byref(self._ctx), int(device),
                                        D, H, W, fh, fw))
        for key, value in (options or {}).items():
            self.set_option(key, value)

    # -- lifetime ---------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self._lib.d3d_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- plumbing ---------------------------------------------------------
    def set_option(self, key, value):
        """Per-context kernel selection switch (include/deconv3d_hip.h: d3d_ctx_set_option)."""
        _check(self._lib.d3d_ctx_set_option(self._ctx, str(key).encode("ascii"), int(value)))

    def get_option(self, key):
        v = C.c_long(0)
        _check(self._lib.d3d_ctx_get_option(self._ctx, str(key).encode("ascii"), C.byref(v)))
        return v.value

    def set_stream(self, stream_handle):
        _check(self._lib.d3d_ctx_set_stream(self._ctx, C.c_void_p(stream_handle)))

    def sync(self):
        _check(self._lib.d3d_sync(self._ctx))

    def timer_start(self):
        _check(self._lib.d3d_timer_start(self._ctx))

    def timer_stop(self):
        ms = C.c_double(0.)
        _check(self._lib.d3d_timer_stop(self._ctx, C.byref(ms)))
        return ms.value

    # -- inputs -----------------------------------------------------------
    def set_taps(self, fsf, lsf, lsf_rel_threshold=-1e-16):
        """lsf_rel_threshold < 0: taps are dropped within that error bound of sum|lsf|
        (include/deconv3d_hip.h); >= 0: relative to the largest tap."""
        fsf = _c64(fsf, self.fsf_shape)
        lsf_p = None
        if lsf is not None:
            lsf = _c64(lsf, (self.shape[0],))
            lsf_p = _dp(lsf)
        _check(self._lib.d3d_set_taps(self._ctx, _dp(fsf), lsf_p,
                                      float(lsf_rel_threshold)))

    def set_data(self, data, var=None, var_scalar=1.0, mask=None):
        data = _c64(data, self.shape)
        var_p = None
        if var is not None:
            var = _c64(var, self.shape)
            var_p = _dp(var)
        mask_p = None
        if mask is not None:
            mask = np.ascontiguousarray(np.asarray(mask) == 1, dtype=np.uint8)
            if mask.shape != self.shape[1:]:
                raise ValueError("mask shape %s != %s" % (mask.shape, self.shape[1:]))
            mask_p = mask.ctypes.data_as(C.POINTER(C.c_uint8))
        _check(self._lib.d3d_set_data(self._ctx, _dp(data), var_p,
                                      float(var_scalar), mask_p))

    def set_params(self, params):
        params = _c64(params, self.shape[1:] + (3,))
        _check(self._lib.d3d_set_params(self._ctx, _dp(params)))

    def get_params(self):
        out = np.empty(self.shape[1:] + (3,), dtype=np.float64)
        _check(self._lib.d3d_get_params(self._ctx, _dp(out)))
        return out

    # -- forward model ----------------------------------------------------
    def build_clean(self):
        out = np.empty(self.shape, dtype=np.float64)
        _check(self._lib.d3d_build_clean(self._ctx, _dp(out)))
        return out

    def convolve(self, cube):
        cube = _c64(cube, self.shape)
        out = np.empty(self.shape, dtype=np.float64)
        _check(self._lib.d3d_convolve(self._ctx, _dp(cube), _dp(out)))
        return out

    def forward(self, fetch=True):
        out = np.empty(self.shape, dtype=np.float64) if fetch else None
        _check(self._lib.d3d_forward(self._ctx, _dp(out) if fetch else None))
        return out

    def simulate(self, params, convolved=True):
        """simulate_clean / simulate_convolved of an explicit parameter map; the
        chain state on the device is left alone."""
        params = _c64(params, self.shape[1:] + (3,))
        out = np.empty(self.shape, dtype=np.float64)
        _check(self._lib.d3d_simulate(self._ctx, _dp(params), 1 if convolved else 0, _dp(out)))
        return out

    def residual(self, fetch=True):
        out = np.empty(self.shape, dtype=np.float64) if fetch else None
        _check(self._lib.d3d_residual(self._ctx, _dp(out) if fetch else None))
        return out

    def chi2_map(self, fetch=True):
        if not fetch:                       # device only, no wait (timing)
            _check(self._lib.d3d_chi2_map(self._ctx, None, None))
            return None
        out = np.empty(self.shape[1:], dtype=np.float64)
        tot = C.c_double(0.)
        _check(self._lib.d3d_chi2_map(self._ctx, _dp(out), C.byref(tot)))
        return out, tot.value

    def upload_slot(self, slot, cube):
        cube = _c64(cube, self.shape)
        _check(self._lib.d3d_upload_slot(self._ctx, int(slot), _dp(cube)))

    def download_slot(self, slot):
        out = np.empty(self.shape, dtype=np.float64)
        _check(self._lib.d3d_download_slot(self._ctx, int(slot), _dp(out)))
        return out

    def convolve_slots(self, src, dst):
        _check(self._lib.d3d_convolve_slots(self._ctx, int(src), int(dst)))

    def stage_upload(self, cube):
        cube = _c64(cube, self.shape)
        _check(self._lib.d3d_stage_upload(self._ctx, _dp(cube)))

    def stage_convolve(self):
        _check(self._lib.d3d_stage_convolve(self._ctx))

    def stage_download(self):
        out = np.empty(self.shape, dtype=np.float64)
        _check(self._lib.d3d_stage_download(self._ctx, _dp(out)))
        return out

    # -- MH within Gibbs --------------------------------------------------
    def mh_config(self, min_b, max_b, jump_amp, gibbs_apriori_variance, seed=12345,
                  refresh_every=1000):
        mn = _c64(min_b, (3,))
        mx = _c64(max_b, (3,))
        amp = _c64(np.ones(3) * np.asarray(jump_amp, dtype=np.float64), (3,))
        _check(self._lib.d3d_mh_config(self._ctx, _dp(mn), _dp(mx), _dp(amp),
                                       float(gibbs_apriori_variance),
                                       C.c_uint64(int(seed) & (2 ** 64 - 1)),
                                       int(refresh_every)))

    def set_sweep_origin(self, origin):
        _check(self._lib.d3d_mh_set_sweep_origin(self._ctx, int(origin)))

    def window_stats(self, y, x, p_new):
        p = _c64(p_new, (3,))
        out = np.empty(5, dtype=np.float64)
        _check(self._lib.d3d_window_stats(self._ctx, int(y), int(x), _dp(p), _dp(out)))
        return out

    def mh_sweeps(self, n_sweeps, first_sweep, keep_one_in=1, chain=None, dlog=None):
        """Returns the number of accepted MH proposals."""
        H, W = self.shape[1:]
        chain_p = None
        dlog_p = None
        last = (first_sweep + n_sweeps - 1) // keep_one_in
        if chain is not None:
            if (chain.dtype != np.float64 or not chain.flags.c_contiguous
                    or chain.shape[1:] != (H, W, 3) or chain.shape[0] <= last):
                raise ValueError("chain must be C-contiguous float64 (n>%d,%d,%d,3)"
                                 % (last, H, W))
            chain_p = _dp(chain)
        if dlog is not None:
            if (dlog.dtype != np.float64 or not dlog.flags.c_contiguous
                    or dlog.shape[1:] != (H, W) or dlog.shape[0] <= last):
                raise ValueError("dlog must be C-contiguous float64 (n>%d,%d,%d)"
                                 % (last, H, W))
            dlog_p = _dp(dlog)
        acc = C.c_int64(0)
        _check(self._lib.d3d_mh_sweeps(self._ctx, int(n_sweeps), int(first_sweep),
                                       int(keep_one_in), chain_p, dlog_p,
                                       C.byref(acc)))
        return acc.value

    def mh_colour_lines(self, sweep, spaxels, in3, lines, gibbs=True):
        """One colour class with host-evaluated unit lines (custom LineModel).
        spaxels [n] local indices, in3 [n,3] = (amplitude, oob, log u),
        lines [n,2,D] = (current, proposed).  Returns [n,3] = (accepted,
        amplitude, delta)."""
        spaxels = np.ascontiguousarray(spaxels, dtype=np.int32)
        n = spaxels.shape[0]
        in3 = _c64(in3, (n, 3))
        lines = _c64(lines, (n, 2, self.shape[0]))
        out = np.empty((n, 3), dtype=np.float64)
        if n:
            _check(self._lib.d3d_mh_colour_lines(
                self._ctx, int(sweep), n, spaxels.ctypes.data_as(C.POINTER(C.c_int)),
                _dp(in3), _dp(lines), 1 if gibbs else 0, _dp(out)))
        return out

    def get_dlog(self):
        out = np.empty(self.shape[1:], dtype=np.float64)
        _check(self._lib.d3d_get_dlog(self._ctx, _dp(out)))
        return out

    def variance_is_uniform(self):
        flag = C.c_int(0)
        _check(self._lib.d3d_variance_is_uniform(self._ctx, C.byref(flag)))
        return bool(flag.value)

    def mh_layers(self):
        n = C.c_int(0)
        _check(self._lib.d3d_mh_layers(self._ctx, C.byref(n)))
        return n.value

    def colour_count(self, colour):
        n = C.c_int(0)
        _check(self._lib.d3d_colour_count(self._ctx, int(colour), C.byref(n)))
        return n.value

    def rtnorm(self, lo, hi, mu=0., sigma=1., size=1, seed=12345, wave_mode=False):
        """`size` draws of N(mu, sigma^2) truncated to [lo, hi] on the device:
        drop-in for rtnorm(a, b, mu, sigma, size) of lib/rtnorm.py:21-92."""
        out = np.empty(int(size), dtype=np.float64)
        _check(self._lib.d3d_rtnorm(self._ctx, int(size), float(lo), float(hi), float(mu),
                                    float(sigma), C.c_uint64(int(seed) & (2 ** 64 - 1)),
                                    1 if wave_mode else 0, _dp(out)))
        return out

    # -- spatial tiling (deconv3d_amd/tiling.py) ----------------------------
    def set_tile(self, gy0, gx0, Wg, oy0, oy1, ox0, ox1):
        _check(self._lib.d3d_set_tile(self._ctx, int(gy0), int(gx0), int(Wg), int(oy0),
                                      int(oy1), int(ox0), int(ox1)))

    def set_parts(self, rects, phases):
        """rects [n,4] = (y0, y1, x0, x1) local, phases [n]."""
        rects = np.ascontiguousarray(rects, dtype=np.int32).reshape(-1, 4)
        phases = np.ascontiguousarray(phases, dtype=np.int32).reshape(-1)
        if rects.shape[0] != phases.shape[0]:
            raise ValueError("one phase per part")
        ip = C.POINTER(C.c_int)
        _check(self._lib.d3d_set_parts(self._ctx, rects.shape[0], rects.ctypes.data_as(ip),
                                       phases.ctypes.data_as(ip)))

    def mh_phase(self, phase, sweep):
        _check(self._lib.d3d_mh_phase(self._ctx, int(phase), int(sweep)))

    def mh_accepted(self, reset=False):
        n = C.c_int64(0)
        _check(self._lib.d3d_mh_accepted(self._ctx, C.byref(n), 1 if reset else 0))
        return n.value

    def flush(self):
        _check(self._lib.d3d_flush(self._ctx))

    def halo_plan(self, plan, entries):
        """entries [n,10] = (peer, kind, send y0,y1,x0,x1, recv y0,y1,x0,x1), local."""
        entries = np.ascontiguousarray(entries, dtype=np.int32).reshape(-1, 10)
        _check(self._lib.d3d_halo_plan(self._ctx, int(plan), entries.shape[0],
                                       entries.ctypes.data_as(C.POINTER(C.c_int))))

    def comm_init(self, nranks, rank, uid):
        if len(uid) != COMM_UID_BYTES:
            raise ValueError("uid must be %d bytes" % COMM_UID_BYTES)
        buf = C.create_string_buffer(bytes(uid), COMM_UID_BYTES)
        _check(self._lib.d3d_comm_init(self._ctx, int(nranks), int(rank), C.cast(buf, C.c_void_p)))

    def comm_destroy(self):
        _check(self._lib.d3d_comm_destroy(self._ctx))

    def comm_info(self):
        """(ranks RCCL says joined the communicator, this rank as RCCL numbers it)."""
        n, r = C.c_int(0), C.c_int(0)
        _check(self._lib.d3d_comm_info(self._ctx, C.byref(n), C.byref(r)))
        return n.value, r.value

    def halo_time(self, reset=False):
        """(milliseconds, exchanges) of the halo exchanges d3d_mh_sweeps ran since the last
        reset (needs option halo_timing = 1)."""
        ms, n = C.c_double(0.), C.c_long(0)
        _check(self._lib.d3d_halo_time(self._ctx, C.byref(ms), C.byref(n), 1 if reset else 0))
        return ms.value, n.value

    def halo_exchange(self, plan):
        _check(self._lib.d3d_halo_exchange(self._ctx, int(plan)))

    def halo_pack(self, plan):
        _check(self._lib.d3d_halo_pack(self._ctx, int(plan)))

    def halo_unpack(self, plan):
        _check(self._lib.d3d_halo_unpack(self._ctx, int(plan)))

    def halo_buffers(self, plan, entry):
        """(send pointer, send bytes, receive pointer, receive bytes) of one entry."""
        sp, rp = C.c_void_p(None), C.c_void_p(None)
        sb, rb = C.c_size_t(0), C.c_size_t(0)
        _check(self._lib.d3d_halo_buffers(self._ctx, int(plan), int(entry), C.byref(sp),
                                          C.byref(sb), C.byref(rp), C.byref(rb)))
        return sp.value, sb.value, rp.value, rb.value

    def halo_download(self, plan, entry):
        _, nbytes, _, _ = self.halo_buffers(plan, entry)
        out = np.empty(nbytes // 8, dtype=np.float64)
        _check(self._lib.d3d_halo_download(self._ctx, int(plan), int(entry), _dp(out)))
        return out

    def halo_upload(self, plan, entry, values):
        _, _, _, nbytes = self.halo_buffers(plan, entry)
        values = np.ascontiguousarray(values, dtype=np.float64).reshape(-1)
        if values.shape[0] * 8 != nbytes:
            raise ValueError("entry expects %d doubles, got %d" % (nbytes // 8, values.shape[0]))
        _check(self._lib.d3d_halo_upload(self._ctx, int(plan), int(entry), _dp(values)))

    def device_copy(self, dst, src, nbytes):
        _check(self._lib.d3d_device_copy(self._ctx, C.c_void_p(dst), C.c_void_p(src),
                                         C.c_size_t(nbytes)))

    def mh_colour(self, colour, sweep):
        _check(self._lib.d3d_mh_colour(self._ctx, int(colour), int(sweep)))

    def export_updates(self, spaxels):
        """[n,8] records {global y, global x, a,c,w before, a,c,w after} of the
        last update of the given local spaxel indices (y*W+x)."""
        spaxels = np.ascontiguousarray(spaxels, dtype=np.int32)
        out = np.empty((spaxels.shape[0], 8), dtype=np.float64)
        if spaxels.shape[0]:
            _check(self._lib.d3d_export_updates(
                self._ctx, spaxels.shape[0],
                spaxels.ctypes.data_as(C.POINTER(C.c_int)), _dp(out)))
        return out

    def apply_updates(self, records):
        """Replay records whose first two columns are LOCAL coordinates."""
        records = np.ascontiguousarray(records, dtype=np.float64).reshape(-1, 8)
        if records.shape[0]:
            _check(self._lib.d3d_apply_updates(self._ctx, records.shape[0], _dp(records)))
