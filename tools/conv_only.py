"""Time only the separable convolution (also the target of rocprofv3 counter
passes):  python tools/conv_only.py [iters] [moffat|gaussian]  -> us per
convolution, spectral + spatial pass between two spectrum-contiguous slots."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, build_taps  # noqa: E402
from deconv3d_amd import _lib  # noqa: E402
from deconv3d_amd.spread_functions import gaussian_image  # noqa: E402

D, H, W, fs = WORKLOADS["c3_300x300x128"]
fsf, lsf = build_taps(D, fs)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
if len(sys.argv) > 2 and sys.argv[2] == "gaussian":
    fsf = gaussian_image(4.0)               # 11x11, an outer product: k_spatial_sep
eng = _lib.Engine((D, H, W), fsf.shape)
eng.set_taps(fsf, lsf)
rng = np.random.default_rng(0)
eng.upload_slot(_lib.SLOT_DATA, rng.normal(size=(D, H, W)))
eng.convolve_slots(_lib.SLOT_DATA, _lib.SLOT_SIM)
eng.sync()
eng.timer_start()
for _ in range(n):
    eng.convolve_slots(_lib.SLOT_DATA, _lib.SLOT_SIM)
ms = eng.timer_stop()
print("%s HY=%s: %.2f us per convolution" % (sys.argv[2] if len(sys.argv) > 2 else "moffat",
                                             os.environ.get("D3D_MARCH_HY", "default"),
                                             ms * 1e3 / n))
eng.close()
