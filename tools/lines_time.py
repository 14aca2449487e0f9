"""Line-cube kernel (lib/line_models.py:92-109 + the LSF of lib/run.py:1011-1024) by option
lines_dense = 0 (tap-list kernel) / 1 (default: k_lines_dense where the line kernel applies the
LSF) / 3 (k_lines_dense always) / 2 (always, own exp), as the forward model's first launch:

    python tools/lines_time.py [DxHxW ...] [rounds=N]

prints us per forward model (k_lines* + the FSF pass), HIP-event timed on the context's stream;
tools/lines_profile.sh runs it under rocprofv3 for the kernels' own durations."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
from deconv3d_amd import _lib  # noqa: E402

rounds = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("rounds=")]
shapes = [a for a in sys.argv[1:] if "=" not in a] or ["128x300x300", "256x300x300", "100x300x300", "64x300x300"]
for spec in shapes:
    D, H, W = [int(v) for v in spec.split("x")]
    fsf, lsf = B.build_taps(D, 11)
    for dense in (0, 1, 3, 2):
        with _lib.Engine((D, H, W), fsf.shape, options={"lines_dense": dense, "lines_rounds": rounds[0] if rounds else 0}) as eng:
            eng.set_taps(fsf, lsf)
            data, var, truth, init, min_b, max_b = B.synthetic_inputs(eng, D, H, W, fsf, 12345)
            eng.set_params(init)
            for _ in range(3):
                eng.forward(fetch=False)
            eng.sync()
            eng.timer_start()
            for _ in range(20):
                eng.forward(fetch=False)
            us = eng.timer_stop() / 20 * 1e3
            print("%s lines_dense=%d: forward model %.1f us" % (spec, dense, us), flush=True)
