"""LSF + FSF convolution of one cube (slot layout) at the shapes round 4 opened to k_conv_rows --
15 x 15 FSFs, depths below 128 that are no power of two -- with the one-pass kernel and with the
kernels it replaces there (option conv_zb = 0 / conv_rows = 0): profiles/r04_conv_shapes.txt.
Run on the GPU box:  python tools/conv_shapes.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import bench as B
from deconv3d_amd import _lib
from deconv3d_amd.spread_functions import moffat_image, muse_like_lsf_vector

CASES = [(128, 300, 300, 15), (128, 300, 300, 13), (128, 300, 300, 11), (256, 300, 300, 15),
         (21, 30, 24, 15), (30, 300, 300, 11), (48, 300, 300, 11), (100, 300, 300, 11), (100, 300, 300, 15)]
from deconv3d_amd.spread_functions import gaussian_image
CASES.append((128, 300, 300, -11))      # a rotated elliptical Gaussian (pa = 30 deg, ba = 0.7): point symmetry only
for D, H, W, fs in CASES:
    if fs < 0:
        fsf = gaussian_image(4.0, pa=30., ba=0.7)
        fs = fsf.shape[0]
    else:
        fsf = moffat_image((fs, fs), beta=2.5, fwhm_px=3.0 if fs <= 11 else 4.0)
    lsf = muse_like_lsf_vector(D, sigma_px=0.9, box_px=1.0)
    for opts in ({}, {"conv_zb": 0} if D != 128 else {"conv_rows": 0}):
        with _lib.Engine((D, H, W), fsf.shape, options=opts) as eng:
            eng.set_taps(fsf, lsf)
            eng.upload_slot(_lib.SLOT_TMP0, np.ones((D, H, W)))
            for it in range(3):
                eng.convolve_slots(_lib.SLOT_TMP0, _lib.SLOT_SIM)
            eng.sync()
            eng.timer_start()
            for it in range(10):
                eng.convolve_slots(_lib.SLOT_TMP0, _lib.SLOT_SIM)
            us = eng.timer_stop() * 100.0
            mb = 2 * D * H * W * 8 / 1e6
            print("D %4d %dx%d FSF %dx%d %-16s: %7.1f us per cube, %5.1f MB algorithmic -> %.3f of the 8 TB/s "
                  "HBM peak" % (D, H, W, fs, fs, opts or "default", us, mb, mb * 1e6 / (us * 1e-6) / 8e12), flush=True)
