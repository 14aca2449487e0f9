"""Run only the separable convolution (for rocprofv3 counter passes)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deconv3d_amd import _lib
from bench import build_taps, WORKLOADS

D, H, W, fs = WORKLOADS["c3_300x300x128"]
fsf, lsf = build_taps(D, fs)
eng = _lib.Engine((D, H, W), fsf.shape)
eng.set_taps(fsf, lsf)
rng = np.random.default_rng(0)
eng.upload_slot(_lib.SLOT_DATA, rng.normal(size=(D, H, W)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for _ in range(n):
    eng.convolve_slots(_lib.SLOT_DATA, _lib.SLOT_SIM)
eng.sync()
eng.close()
