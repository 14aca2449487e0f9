#!/usr/bin/env python
# coding=utf-8
"""
Anatomy of a colour launch that does NOT fill the chip (needs `make EXPERIMENTS=1` for the
stamps; the timing line works with any build).

    python tools/mh_tail.py DxHxW[,fs] [key=value ...]      e.g.  64x64x64  mh_small=0
                                                                   128x37x300 parts=1

Per launch (median over the launches of one sweep), 100 MHz wall-clock stamps relative to the
first workgroup's entry: setup done, window streamed (median / last), prepare wavefront done,
channel sums in LDS, verdict in LDS, update written, next launch's first entry -- and WHERE the
workgroups ran (HW_ID / XCC_ID): distinct compute units, workgroups that shared one, and the
stream time by that.  `parts=1` partitions the cube like an 8x1 tile (d3d_set_parts: far part +
near part), which selects the wide form at 128 channels.
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from deconv3d_amd import _lib  # noqa: E402


def make_engine(D, H, W, fs, opts, parts):
    fsf, lsf = bench.build_taps(D, fs)
    eng = _lib.Engine((D, H, W), fsf.shape, options=opts)
    eng.set_taps(fsf, lsf)
    data, var, truth, init, min_b, max_b = bench.synthetic_inputs(eng, D, H, W, fsf, 12345)
    eng.set_data(data, var)
    if parts:
        fhh = (fsf.shape[0] - 1) // 2
        near = 2 * fhh
        eng.set_parts([(0, H - near, 0, W), (H - near, H, 0, W)], [0, 1])
    eng.set_params(init)
    eng.mh_config(min_b, max_b, 0.1, float(max_b[0] ** 2), seed=1, refresh_every=0)
    eng.residual(fetch=False)
    return eng, fsf


def main():
    shape = sys.argv[1].split(",")
    D, H, W = (int(v) for v in shape[0].split("x"))
    fs = int(shape[1]) if len(shape) > 1 else 11
    opts, parts = {}, False
    for kv in sys.argv[2:]:
        k, v = kv.split("=")
        if k == "parts":
            parts = bool(int(v))
        else:
            opts[k] = int(v)
    eng, fsf = make_engine(D, H, W, fs, opts, parts)
    ncol = fsf.size
    # plain timing first (no stamps armed)
    eng.mh_sweeps(3, 1)
    eng.sync()
    eng.timer_start()
    n = 20
    eng.mh_sweeps(n, 4)
    ms = eng.timer_stop()
    nparts = 2 if parts else 1
    print("%s fs=%d %s parts=%d: %.4f ms per sweep, %.2f us per colour launch, %.3f M updates/s"
          % (sys.argv[1], fs, opts, nparts, ms / n, ms * 1e3 / (n * ncol * nparts),
             H * W * n / ms / 1e3))
    lib = eng._lib
    if not hasattr(lib, "d3d_x_stamps_arm"):
        eng.close()
        return
    lib.d3d_x_stamps_arm.argtypes = [C.c_void_p, C.c_int]
    lib.d3d_x_stamps_read.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
    nl = 2 * ncol * nparts
    assert lib.d3d_x_stamps_arm(eng._ctx, nl) == 0
    eng.mh_sweeps(2, 30)
    cap = (H // fsf.shape[0] + 3) * (W // fsf.shape[1] + 3)
    rows = []
    for launch in range(nl):
        buf = np.zeros(cap * 8, dtype=np.uint64)
        assert lib.d3d_x_stamps_read(eng._ctx, launch, cap, buf.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
        st = buf.reshape(cap, 8)
        rows.append(st[st[:, 0] > 0])
    names = ["workgroups", "real", "setup done (med)", "stream done (med)", "stream done (last)",
             "prepare done (med)", "sums in LDS (last)", "verdict in LDS (last)", "update written (last)",
             "next launch starts", "distinct CUs", "most WGs on one CU"]
    t, dur_by_share = [], {}
    half = nl // 2
    for k in range(half, nl - 1):
        st, nxt = rows[k], rows[k + 1]
        if len(st) == 0 or len(nxt) == 0:
            continue
        f = st.astype(np.float64)
        real = st[:, 4] > 0
        if not real.any():
            continue
        t0 = f[:, 0].min()
        us = lambda v: (v - t0) / 100.0
        hw = st[:, 5]
        cu = ((hw >> np.uint64(32)) << np.uint64(16)) | (hw & np.uint64(0xff00))  # XCC | SE, SH, CU bits
        streaming = st[:, 2] > 0
        uniq, cnt = np.unique(cu[streaming], return_counts=True)
        share = dict(zip(uniq.tolist(), cnt.tolist()))
        for i in np.nonzero(real)[0]:
            dur_by_share.setdefault(share[int(cu[i])], []).append((f[i, 2] - f[i, 1]) / 100.0)
        t.append([len(st), real.sum(), np.median(us(f[:, 1])), np.median(us(f[streaming, 2])),
                  us(f[streaming, 2].max()), np.median(us(f[real, 3])) if (st[real, 3] > 0).any() else np.nan,
                  us(f[real, 6].max()), us(f[real, 7].max()), us(f[real, 4].max()), us(nxt[:, 0].astype(np.float64).min()),
                  len(uniq), cnt.max()])
    t = np.array(t)
    print("  us from the first workgroup's entry, median over %d launches" % len(t))
    for i, nm in enumerate(names):
        print("  %-26s %8.2f   (min %.2f, max %.2f)" % (nm, np.nanmedian(t[:, i]), np.nanmin(t[:, i]),
                                                        np.nanmax(t[:, i])))
    for sh in sorted(dur_by_share):
        v = np.array(dur_by_share[sh])
        print("  real windows on a CU shared by %d streaming workgroup(s): n %d, stream p50 %.2f p95 %.2f us"
              % (sh, len(v), np.median(v), np.percentile(v, 95)))
    eng.close()


if __name__ == "__main__":
    main()
