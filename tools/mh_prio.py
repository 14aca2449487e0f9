#!/usr/bin/env python
# coding=utf-8
"""
Staggered completion of a colour's windows (option mh_prio: 1..15 -- the workgroups with
bit mh_prio-1 of their index set run at s_setprio 2; 16 + n -- the odd workgroups start
n x 0.8 us late), for the per-colour
launches and -- in an EXPERIMENTS build -- for two colour classes per launch (k_mh_pair),
on BASELINE config 3.  Prints microseconds per colour class.

    python tools/mh_prio.py [sweeps]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
from deconv3d_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
D, H, W, fs = B.WORKLOADS["c3_300x300x128"]
fsf, lsf = B.build_taps(D, fs)
with _lib.Engine((D, H, W), fsf.shape) as e0:
    e0.set_taps(fsf, lsf)
    data, var, truth, init, min_b, max_b = B.synthetic_inputs(e0, D, H, W, fsf, 12345)
variants = [{"mh_prio": p} for p in (0, 1, 2, 4, 9, 18, 21, 26)]       # 16 + n: odd workgroups start n x 0.8 us late
if _lib.has_experiments():
    variants += [{"mh_pair": 1, "mh_prio": p} for p in (0, 1, 4, 18, 21, 26)]
ref = None
for rep in range(2):
    for opts in variants:
        with _lib.Engine((D, H, W), fsf.shape, options=opts) as eng:
            eng.set_taps(fsf, lsf)
            eng.set_data(data, var)
            eng.set_params(init)
            eng.mh_config(min_b, max_b, 0.1, float(max_b[0] ** 2), seed=12345, refresh_every=0)
            eng.residual(fetch=False)
            eng.mh_sweeps(5, 1)
            eng.sync()
            eng.timer_start()
            eng.mh_sweeps(n, 6)
            ms = eng.timer_stop()
            p = eng.get_params()
            if ref is None:
                ref = p
            same = bool(np.array_equal(p, ref))
            print("%-32s %.2f us per colour class (%.3f ms per sweep)  chain bit-identical to mh_prio=0: %s"
                  % (opts, ms * 1e3 / n / (fs * fs), ms / n, same), flush=True)
