"""LSF + FSF convolution of one cube in the slot layout at several depths, with and without
the z-blocked k_conv_rows (option conv_zb): profiles/r03_conv_depths.txt.
Run on the GPU box:  python tools/conv_depths.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np

import bench as B
from deconv3d_amd import _lib

for D in (100, 128, 200, 256, 512, 1024, 3682):
    H = W = 300 if D <= 512 else (150 if D <= 1024 else 60)
    fsf, lsf = B.build_taps(D, 11)
    for zb in (0, 1):
        with _lib.Engine((D, H, W), fsf.shape, options={"conv_zb": zb}) as eng:
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
            print("D %4d %dx%d conv_zb %d: %7.1f us per cube, %4.0f MB algorithmic -> %.3f of the 8 TB/s "
                  "HBM peak" % (D, H, W, zb, us, mb, mb * 1e6 / (us * 1e-6) / 8e12), flush=True)
