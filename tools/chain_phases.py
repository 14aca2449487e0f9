#!/usr/bin/env python
# coding=utf-8
"""
Where the time of one colour of k_mh_chain goes (needs a `make EXPERIMENTS=1` build:
DECONV3D_HIP_LIB=deconv3d_amd/csrc/exp/libdeconv3d_hip.so).

    python tools/chain_phases.py [workload | DxHxW] [--part TYxTX]

Arms the phase stamps (100 MHz wall clock; per slot and colour, by thread 0: colour starts, entering
columns' loads issued (predecessors' residual stores seen), predecessors' G rows staged,
barrier, window consumed, barrier, channel sums done, decision taken / next colour prepared),
runs sweeps of the workload (or of the FF part of an interior rank of a TYxTX tiling of it)
and prints the median duration of each phase over slots and colours of the last sweep, and
the colour-to-colour period.
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from deconv3d_amd import _lib, tiling  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("workload", nargs="?", default="c2_64x64x64")
ap.add_argument("--part", default=None)
args = ap.parse_args()
if args.workload in bench.WORKLOADS:
    D, H, W, fs = bench.WORKLOADS[args.workload]
else:
    D, H, W = [int(v) for v in args.workload.split("x")]
    fs = 11
fsf, lsf = bench.build_taps(D, fs)
eng = _lib.Engine((D, H, W), fsf.shape)
eng.set_taps(fsf, lsf)
data, var, truth, init, min_b, max_b = bench.synthetic_inputs(eng, D, H, W, fsf, 12345)
ra = float(max_b[0] ** 2)
if args.part:
    ty, tx = [int(v) for v in args.part.split("x")]
    lay = tiling.TileLayout(H, W, fs, fs, ty, tx)
    rank = (ty // 2) * tx + tx // 2 if ty * tx > 2 else 0
    eng.close()
    eng = tiling.make_tile_engine(lay, rank, data, var, np.ones((H, W)), fsf, lsf, init, min_b,
                                  max_b, 0.1, ra, 12345)
    phases = lay.phases
else:
    eng.set_data(data, var)
    eng.set_params(init)
    eng.mh_config(min_b, max_b, 0.1, ra, seed=1, refresh_every=0)
    eng.residual(fetch=False)
    phases = None
lib = eng._lib
if not hasattr(lib, "d3d_x_stamps_arm"):
    raise SystemExit("library built without EXPERIMENTS=1")
assert eng.get_option("chain_parts") >= 1, "no part of this context takes the chain form"
K = fs * fs
lib.d3d_x_stamps_arm.argtypes = [C.c_void_p, C.c_int]
lib.d3d_x_stamps_raw.argtypes = [C.c_void_p, C.c_long, C.c_long, C.POINTER(C.c_uint64)]


def sweep(s, n=1):
    if phases is None:
        eng.mh_sweeps(n, s)
    else:
        for k in range(n):
            eng.mh_phase(phases[0], s + k)     # the FF part only


sweep(1, 2)
assert lib.d3d_x_stamps_arm(eng._ctx, 2 * K) == 0
sweep(3, 1)
eng.sync()
slots = 1024
buf = np.zeros(slots * K * 8, dtype=np.uint64)
n = 0
for trial in (1024, 512, 256, 128, 64, 32, 16):
    if lib.d3d_x_stamps_raw(eng._ctx, 0, trial * K * 8, buf.ctypes.data_as(C.POINTER(C.c_uint64))) == 0:
        n = trial
        break
st = buf[:n * K * 8].reshape(n, K, 8).astype(np.float64)
live = st[:, :, 0] > 0
st = st[live.all(axis=1)]
print("%dx%dx%d%s: %d slots stamped, %d colours" % (D, H, W, " part of " + args.part if args.part else "",
                                                    st.shape[0], K))
names = ["flags of remote predecessors, G rows, entering loads", "-", "-",
         "apply + store + accumulate (wavefront 0)", "barrier B1", "channel sums + B2",
         "decision | drain, flag1, next lines + B3"]
d = np.diff(st[:, :, :8], axis=2) / 100.0          # us
tot = 0.0
for j, nm in enumerate(names):
    print("  %-52s median %6.2f us   p90 %6.2f" % (nm, np.median(d[:, 1:, j]), np.percentile(d[:, 1:, j], 90)))
    tot += np.median(d[:, 1:, j])
gap = (st[:, 1:, 0] - st[:, :-1, 7]) / 100.0
print("  %-52s median %6.2f us" % ("G row to LDS + memory, B4 -> next colour starts", np.median(gap)))
period = (st[:, 1:, 0] - st[:, :-1, 0]) / 100.0
print("  colour period: median %.2f us (sum of medians %.2f); sweep span %.3f ms"
      % (np.median(period), tot + np.median(gap), (st[:, -1, 7].max() - st[:, 0, 0].min()) / 1e5))
