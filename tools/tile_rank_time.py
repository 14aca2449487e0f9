"""What ONE rank of a tiled chain computes per sweep, measured alone on one GPU (its phases
back to back, no halo traffic), beside the whole cube's sweep on the same GPU in the same
run: the compute side of the strong-scaling projection of DESIGN.md section 7.

    python tools/tile_rank_time.py [--hw 300x300] [--depth 128] [TYxTX ...]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
from deconv3d_amd import _lib, tiling  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--hw", default="300x300")
ap.add_argument("--depth", type=int, default=128)
ap.add_argument("--sweeps", type=int, default=10)
ap.add_argument("layouts", nargs="*", default=["2x1", "4x1", "8x1", "2x2", "2x4"])
args = ap.parse_args()
H, W = [int(v) for v in args.hw.split("x")]
D, fs = args.depth, 11
fsf, lsf = B.build_taps(D, fs)
n = args.sweeps

with _lib.Engine((D, H, W), fsf.shape) as full:
    full.set_taps(fsf, lsf)
    data, var, truth, init, min_b, max_b = B.synthetic_inputs(full, D, H, W, fsf, 12345)
    ra = float(max_b[0] ** 2)
    full.set_data(data, var, mask=None)
    full.set_params(init)
    full.mh_config(min_b, max_b, 0.1, ra, seed=12345, refresh_every=0)
    full.residual(fetch=False)
    full.mh_sweeps(2, 1)
    full.sync()
    t0 = time.perf_counter()
    full.mh_sweeps(n, 3)
    full.sync()
    one = (time.perf_counter() - t0) * 1e3 / n
print("%dx%dx%d, one GPU, whole cube: %.3f ms per sweep (%.2f M spaxel-updates/s)"
      % (H, W, D, one, H * W / one / 1e3), flush=True)

mask = np.ones((H, W))
for spec in args.layouts:
    ty, tx = [int(v) for v in spec.split("x")]
    lay = tiling.TileLayout(H, W, fs, fs, ty, tx)
    rank = (ty // 2) * tx + tx // 2 if ty * tx > 2 else 0          # an interior rank where there is one
    eng = tiling.make_tile_engine(lay, rank, data, var, mask, fsf, lsf, init, min_b, max_b, 0.1, ra,
                                  12345)
    for s in range(1, 3):
        for ph in lay.phases:
            eng.mh_phase(ph, s)
    eng.sync()
    t0 = time.perf_counter()
    for s in range(3, 3 + n):
        for ph in lay.phases:
            eng.mh_phase(ph, s)
    eng.sync()
    ms = (time.perf_counter() - t0) * 1e3 / n
    parts = ", ".join("%s %dx%d" % ("FF FN NF NN".split()[ph], r[1] - r[0], r[3] - r[2])
                      for ph, r in lay.parts(rank))
    halo = sum(int((r[3] - r[2]) * (r[5] - r[4])) for ph in lay.phases
               for r in tiling.plan_tables(lay, rank)[ph]) * D * 8
    # one xGMI link per neighbour, ~50 GB/s effective for a 3 MB message each way
    print("%s rank %d: %.3f ms per sweep alone (parts: %s; sends %.1f MB per sweep) -> %.2fx of the "
          "whole cube's %.2f ms before halo time" % (spec, rank, ms, parts, halo / 1e6, one / ms, one),
          flush=True)
    eng.close()
