"""Long-horizon check of the tiled chain at config 4's size on ONE GPU: tile contexts in
loop-back against the partitioned single context over many sweeps, crossing the periodic
from-scratch residual (lib/run.py:521-534) several times; parameters, residual and accepted
counts must stay bit-identical.   python tools/tile_soak.py [--sweeps 260] [--refresh 100] [TYxTX ...]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
from deconv3d_amd import _lib, tiling  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sweeps", type=int, default=260)
ap.add_argument("--refresh", type=int, default=100)
ap.add_argument("layouts", nargs="*", default=["8x1", "2x4"])
args = ap.parse_args()

D, H, W, fs = B.WORKLOADS["c3_300x300x128"]
fsf, lsf = B.build_taps(D, fs)
with _lib.Engine((D, H, W), fsf.shape) as full:
    full.set_taps(fsf, lsf)
    data, var, truth, init, min_b, max_b = B.synthetic_inputs(full, D, H, W, fsf, 12345)
mask = np.ones((H, W))
mask[17, 200] = mask[151, 149] = mask[150, 150] = 0
ra = float(max_b[0] ** 2)
bad = 0
for spec in args.layouts:
    ty, tx = [int(v) for v in spec.split("x")]
    lay = tiling.TileLayout(H, W, fs, fs, ty, tx)
    with _lib.Engine((D, H, W), fsf.shape) as ref:
        ref.set_taps(fsf, lsf)
        ref.set_data(data, var, mask=mask)
        tiling.apply_parts(ref, lay)
        ref.set_params(init)
        ref.mh_config(min_b, max_b, 0.1, ra, seed=12345, refresh_every=args.refresh)
        err0 = ref.residual()
        t0 = time.perf_counter()
        accepted = ref.mh_sweeps(args.sweeps, 1)
        t_ref = time.perf_counter() - t0
        ref_params = ref.get_params()
        ref_err = ref.download_slot(_lib.SLOT_ERR)
    engines = [tiling.make_tile_engine(lay, r, data, var, mask, fsf, lsf, init, min_b, max_b, 0.1, ra,
                                       12345, err=err0, refresh_every=args.refresh) for r in range(lay.n)]
    tables = [tiling.plan_tables(lay, r) for r in range(lay.n)]
    t0 = time.perf_counter()
    for s in range(1, args.sweeps + 1):
        tiling.sweep_loopback(engines, lay, tables, s, device_copy=True, refresh=(s % args.refresh == 0))
    for e in engines:
        e.sync()
    t_tiles = time.perf_counter() - t0
    worst_p = worst_e = 0.0
    for r, e in enumerate(engines):
        (oy0, oy1, ox0, ox1), p = tiling.gather_params(lay, r, e)
        live = mask[oy0:oy1, ox0:ox1] == 1
        worst_p = max(worst_p, float(np.max(np.abs(p[live] - ref_params[oy0:oy1, ox0:ox1][live]))))
        uy0, uy1, ux0, ux1 = lay.used(r)                    # every cell the rank's windows touch
        ry0, _, rx0, _ = lay.region(r)
        err = e.download_slot(_lib.SLOT_ERR)[:, uy0 - ry0:uy1 - ry0, ux0 - rx0:ux1 - rx0]
        worst_e = max(worst_e, float(np.max(np.abs(err - ref_err[:, uy0:uy1, ux0:ux1]))))
    acc = sum(e.mh_accepted() for e in engines)
    ok = worst_p == 0.0 and worst_e == 0.0 and acc == accepted
    bad += not ok
    print("%s: %d sweeps, residual rebuilt every %d: max |d params| %.3g, max |d residual| %.3g, accepted "
          "%d vs %d -> %s   (single context %.1f s, %d tile contexts in loop-back %.1f s)"
          % (spec, args.sweeps, args.refresh, worst_p, worst_e, acc, accepted,
             "BIT-IDENTICAL" if ok else "DIFFERENT", t_ref, lay.n, t_tiles), flush=True)
    for e in engines:
        e.close()
sys.exit(1 if bad else 0)
