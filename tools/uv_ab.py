"""A/B of the uniform-variance variant of the sweep kernel (the reference's default, variance=None:
one constant, lib/run.py:186-192) at 300x300x128: pending layers 2 / 3, as bench.py's
`uniform_variance` leg measures it.   python tools/uv_ab.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import bench as B
from deconv3d_amd import _lib

D, H, W, fs = B.WORKLOADS["c3_300x300x128"]
fsf, lsf = B.build_taps(D, fs)
for opts in ({}, {"mh_layers": 3}, {"mh_layers": 1}, {"uniform_ivar": 0}, {"uniform_ivar": 0, "mh_layers": 3}):
    with _lib.Engine((D, H, W), fsf.shape, options=opts) as eng:
        eng.set_taps(fsf, lsf)
        data, var, truth, init, min_b, max_b = B.synthetic_inputs(eng, D, H, W, fsf, 12345)
        eng.set_data(data, None, var_scalar=float(np.mean(var)))
        eng.set_params(init)
        eng.mh_config(min_b, max_b, 0.1, float(max_b[0] ** 2), seed=12345, refresh_every=0)
        eng.residual(fetch=False)
        eng.mh_sweeps(3, 1)
        eng.sync()
        eng.timer_start()
        eng.mh_sweeps(20, 4)
        ms = eng.timer_stop()
        us = ms * 1e3 / (20 * 121)
        nb = 16 * D * B.window_voxels(H, W, 11, 11) / 121
        print("%-40s %.2f us per launch, %.2f M updates/s, %.3f of the HBM peak at 16 B per window voxel (%s)"
              % (opts or "default", us, 20 * H * W / ms / 1e3, nb / (us * 1e-6) / 8e12,
                 "uniform kernel" if eng.variance_is_uniform() else "general kernel"), flush=True)
