"""MH sweep time of one context for arbitrary cube sizes (per-voxel variance, Moffat 11x11,
17-tap LSF): what the Infinity Cache does and does not hold.
    python tools/mh_sizes.py [DxHxW ...]      (D3D_MH_ZIGZAG=0 for the A/B of DESIGN.md section 3)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
from deconv3d_amd import _lib  # noqa: E402

shapes = sys.argv[1:] or ["128x300x300", "256x300x300", "128x600x600"]
for spec in shapes:
    D, H, W = [int(v) for v in spec.lower().split("x")]
    fsf, lsf = B.build_taps(D, 11)
    with _lib.Engine((D, H, W), fsf.shape) as eng:
        eng.set_taps(fsf, lsf)
        data, var, truth, init, mn, mx = B.synthetic_inputs(eng, D, H, W, fsf, 777)
        eng.set_data(data, var, mask=None)
        del data, var
        eng.set_params(init)
        eng.mh_config(mn, mx, 0.1, float(mx[0] ** 2), seed=777, refresh_every=0)
        eng.residual(fetch=False)
        eng.mh_sweeps(2, 1)
        eng.sync()
        n = 6
        eng.timer_start()
        acc = eng.mh_sweeps(n, 3)
        ms = eng.timer_stop()
        ws = 16.0 * D * H * W / 1e6
        frac = 24.0 * D * B.window_voxels(H, W, 11, 11) / 121 / (ms * 1e-3 / n / 121) / 8e12
        print("zigzag=%s %dx%dx%d (residual + 1/variance %.0f MB; beyond-cache policy %s): %.3f ms per sweep, "
              "%.2f us per launch, %.2f M updates/s, %.3f of the HBM peak, accepted %d"
              % (os.environ.get("D3D_MH_ZIGZAG", "1"), D, H, W, ws,
                 "on" if eng.get_option("mh_nt_ivar_on") else "off", ms / n,
                 ms * 1e3 / n / 121, H * W * n / ms / 1e3, frac, acc), flush=True)
