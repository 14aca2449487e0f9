"""The compute side of a tiled chain's strong scaling, measured on ONE GPU.

    python tools/tile_rank_time.py [--hw 300x300] [--depth 128] [TYxTX ...]

A tiled sweep runs its phases one after the other, every rank at once: its compute time is the
CRITICAL PATH  sum over the phases of the slowest rank in that phase  (tiling.critical_path).
So ONE representative of every class of ranks (tiling.rank_classes: same parts, same launches)
is built as a tile context and each of its phases timed alone on the GPU -- no halo traffic --
beside the whole cube's sweep on the same GPU in the same run.  Printed per layout: every
class's phases, the critical path, the rank and phase it comes from, and whole cube / critical
path = the speed-up the grid can reach before its halo copies.
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

NAMES = "FF FN NF NN".split()
mask = np.ones((H, W))
for spec in args.layouts:
    ty, tx = [int(v) for v in spec.split("x")]
    lay = tiling.TileLayout(H, W, fs, fs, ty, tx)
    timed = {}
    print("%s (%d ranks, phases %s):" % (spec, lay.n, " ".join(NAMES[p] for p in lay.phases)), flush=True)
    for key, ranks in tiling.rank_classes(lay).items():
        rank = ranks[len(ranks) // 2]
        eng = tiling.make_tile_engine(lay, rank, data, var, mask, fsf, lsf, init, min_b, max_b, 0.1, ra,
                                      12345)
        tiling.time_phases(eng, lay, rank, 1, 2)               # warm
        t = tiling.time_phases(eng, lay, rank, 3, n)
        eng.close()
        for r in ranks:
            timed[r] = t
        halo = sum(int((r[3] - r[2]) * (r[5] - r[4])) for ph in lay.phases
                   for r in tiling.plan_tables(lay, rank)[ph]) * D * 8
        print("   ranks %-14s %s  = %.3f ms alone; sends %.1f MB per sweep"
              % (",".join(str(r) for r in ranks),
                 "  ".join("%s %dx%d %.3f ms" % (NAMES[ph], hh, ww, t[ph]) for ph, hh, ww in key),
                 sum(t.values()), halo / 1e6), flush=True)
    cp, by_phase, busiest = tiling.critical_path(timed)
    print("   critical path %.3f ms per sweep = %s  -> %.2fx of the whole cube's %.3f ms before halo "
          "time (busiest rank %d: %.3f ms)"
          % (cp, " + ".join("%s %.3f (rank %d)" % (NAMES[ph], ms, r) for ph, (r, ms) in by_phase.items()),
             one / cp, one, busiest, sum(timed[busiest].values())), flush=True)
