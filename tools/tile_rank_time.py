"""What ONE rank of a tiled 300x300x128 chain computes per sweep, measured alone on one GPU
(its phases back to back, no halo traffic): the compute side of the strong-scaling projection
of DESIGN.md section 7.   python tools/tile_rank_time.py [TYxTX ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
from deconv3d_amd import _lib, tiling  # noqa: E402

D, H, W, fs = B.WORKLOADS["c3_300x300x128"]
fsf, lsf = B.build_taps(D, fs)
with _lib.Engine((D, H, W), fsf.shape) as full:
    full.set_taps(fsf, lsf)
    data, var, truth, init, min_b, max_b = B.synthetic_inputs(full, D, H, W, fsf, 12345)
mask = np.ones((H, W))
ra = float(max_b[0] ** 2)
for spec in (sys.argv[1:] or ["2x1", "4x1", "8x1", "2x2", "2x4"]):
    ty, tx = [int(v) for v in spec.split("x")]
    lay = tiling.TileLayout(H, W, fs, fs, ty, tx)
    rank = (ty // 2) * tx + tx // 2 if ty * tx > 2 else 0          # an interior rank where there is one
    eng = tiling.make_tile_engine(lay, rank, data, var, mask, fsf, lsf, init, min_b, max_b, 0.1, ra,
                                  12345)
    n = 10
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
    print("%s rank %d: %.3f ms per sweep alone (parts: %s; sends %.1f MB per sweep) -> %.2fx of one "
          "GPU's 5.10 ms before halo time" % (spec, rank, ms, parts, halo / 1e6, 5.10 / ms))
    eng.close()
