"""The reference-layout convolution (d3d_stage_convolve: k_spectral_z + k_spatial_z, what
d3d_convolve runs between its upload and its download) by rows per strip (option zmajor_hy):

    python tools/zmajor_time.py [DxHxW] [hy ...]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
from deconv3d_amd import _lib  # noqa: E402

spec = sys.argv[1] if len(sys.argv) > 1 else "128x300x300"
D, H, W = [int(v) for v in spec.split("x")]
hys = [int(v) for v in sys.argv[2:]] or [0, 16, 32, 50, 60, 75, 100, 150, 300]
fsf, lsf = B.build_taps(D, 11)
cube = np.random.default_rng(1).normal(size=(D, H, W))
for hy in hys:
    opts = {"zmajor_hy": hy} if hy else {}
    with _lib.Engine((D, H, W), fsf.shape, options=opts) as eng:
        eng.set_taps(fsf, lsf)
        eng.stage_upload(cube)
        for _ in range(3):
            eng.stage_convolve()
        eng.sync()
        eng.timer_start()
        for _ in range(20):
            eng.stage_convolve()
        ms = eng.timer_stop() / 20
        print("%s zmajor_hy=%s: %.1f us per convolution (%.3f of 8 TB/s for %d MB)"
              % (spec, hy or "default", ms * 1e3, 2 * 8 * D * H * W / (ms * 1e-3) / 8e12, 2 * 8 * D * H * W / 1e6),
              flush=True)
