#!/usr/bin/env python
# coding=utf-8
"""
Where the time of one k_mh_ws launch goes (needs `make EXPERIMENTS=1`).

    python tools/mh_phases.py [uniform|full] [workload]

Arms the phase stamps of the library (100 MHz wall clock, five per workgroup:
entry, setup done, window streamed, prepare wavefront done, update written),
runs two sweeps of the 300x300x128 / 11x11 workload and prints, per launch
(median over the launches of the second sweep), when the phases start and end
relative to the first workgroup's entry, and the gap to the next launch.
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from deconv3d_amd import _lib  # noqa: E402


def main():
    uniform = len(sys.argv) > 1 and sys.argv[1] == "uniform"
    workload = sys.argv[2] if len(sys.argv) > 2 else "c3_300x300x128"
    D, H, W, fs = bench.WORKLOADS[workload]
    fsf, lsf = bench.build_taps(D, fs)
    eng = _lib.Engine((D, H, W), fsf.shape)
    eng.set_taps(fsf, lsf)
    data, var, truth, init, min_b, max_b = bench.synthetic_inputs(eng, D, H, W, fsf, 12345)
    if uniform:
        eng.set_data(data, None, var_scalar=float(var.mean()))
    else:
        eng.set_data(data, var)
    eng.set_params(init)
    eng.mh_config(min_b, max_b, 0.1, float(max_b[0] ** 2), seed=1, refresh_every=0)
    eng.residual(fetch=False)
    eng.mh_sweeps(2, 1)
    lib = eng._lib
    ncol = fsf.size
    if not hasattr(lib, "d3d_x_stamps_arm"):
        raise SystemExit("library built without EXPERIMENTS=1")
    lib.d3d_x_stamps_arm.argtypes = [C.c_void_p, C.c_int]
    lib.d3d_x_stamps_read.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint64)]
    assert lib.d3d_x_stamps_arm(eng._ctx, 2 * ncol) == 0
    eng.mh_sweeps(2, 3)
    rows = []
    for launch in range(2 * ncol):
        n = 0
        # work-list length of this colour: real + virtual positions
        col = launch % ncol
        n = (H // fsf.shape[0] + 3) * (W // fsf.shape[1] + 3)
        buf = np.zeros(n * 8, dtype=np.uint64)
        assert lib.d3d_x_stamps_read(eng._ctx, launch, n, buf.ctypes.data_as(
            C.POINTER(C.c_uint64))) == 0
        st = buf.reshape(n, 8)[:, :5].astype(np.float64)
        st = st[st[:, 0] > 0]
        rows.append(st)
    t = []
    for k in range(ncol, 2 * ncol - 1):
        st, nxt = rows[k], rows[k + 1]
        real = st[:, 4] > 0
        t0 = st[:, 0].min()
        us = lambda v: (v - t0) / 100.0        # 100 MHz -> microseconds
        t.append([len(st), real.sum(),
                  us(st[:, 0].max()),                        # last workgroup enters
                  np.median(us(st[:, 1])),                   # setup done (median)
                  np.median(us(st[:, 2])), us(st[:, 2].max()),   # stream done (median, last)
                  np.median(us(st[real, 3])),                # prepare done
                  us(st[real, 4].max()),                     # last update written
                  us(nxt[:, 0].min())])                      # next launch's first entry
    t = np.array(t)
    # per-workgroup durations over the same launches
    stream, tail, by_xcd = [], [], [[] for _ in range(8)]
    for k in range(ncol, 2 * ncol - 1):
        st = rows[k]
        real = st[:, 4] > 0
        stream.append((st[:, 2] - st[:, 1]) / 100.0)
        tail.append((st[real, 4] - st[real, 2]) / 100.0)
        t0 = st[:, 0].min()
        for xcd in range(8):
            by_xcd[xcd].append(np.median((st[xcd::8, 2] - t0) / 100.0))
    stream, tail = np.concatenate(stream), np.concatenate(tail)
    names = ["workgroups", "real", "last entry", "setup done (med)", "stream done (med)",
             "stream done (last)", "prepare done (med)", "last update written",
             "next launch starts"]
    print("k_mh_ws phases, %s, %s variance, us from the first workgroup's entry "
          "(median over %d launches)" % (workload, "uniform" if uniform else "per-voxel", len(t)))
    for i, nm in enumerate(names):
        print("  %-22s %8.2f   (min %.2f, max %.2f)" % (nm, np.median(t[:, i]), t[:, i].min(),
                                                        t[:, i].max()))
    pct = lambda v: "p5 %.2f  p50 %.2f  p95 %.2f  max %.2f" % tuple(
        np.percentile(v, [5, 50, 95, 100]))
    print("  per workgroup: stream %s" % pct(stream))
    print("  per workgroup: tail   %s" % pct(tail))
    print("  stream done (median) by workgroup index mod 8: " +
          " ".join("%.1f" % np.median(v) for v in by_xcd))
    eng.close()


if __name__ == "__main__":
    main()
