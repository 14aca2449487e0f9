import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.getcwd())
import bench
from deconv3d_amd import _lib
for wl in ("c2_64x64x64", "c1_32x16x16"):
    D, H, W, fs = bench.WORKLOADS[wl]
    fsf, lsf = bench.build_taps(D, fs)
    eng = _lib.Engine((D, H, W), fsf.shape)
    eng.set_taps(fsf, lsf)
    data, var, truth, init, min_b, max_b = bench.synthetic_inputs(eng, D, H, W, fsf, 12345)
    eng.set_data(data, var); eng.set_params(init)
    eng.mh_config(min_b, max_b, 0.1, float(max_b[0] ** 2), seed=1, refresh_every=0)
    eng.residual(fetch=False)
    lib = eng._lib
    K = fs * fs
    lib.d3d_x_stamps_arm.argtypes = [C.c_void_p, C.c_int]
    lib.d3d_x_stamps_raw.argtypes = [C.c_void_p, C.c_long, C.c_long, C.POINTER(C.c_uint64)]
    eng.mh_sweeps(50, 1)
    assert lib.d3d_x_stamps_arm(eng._ctx, 2 * K) == 0
    eng.mh_sweeps(50, 51)
    eng.sync()
    n = 8
    buf = np.zeros(n * K * 8, dtype=np.uint64)
    assert lib.d3d_x_stamps_raw(eng._ctx, 0, n * K * 8, buf.ctypes.data_as(C.POINTER(C.c_uint64))) == 0
    st = buf.reshape(n, K, 8).astype(np.float64)
    real = np.diff(st[:, :, 0], axis=1) / 100.0   # us
    cyc = np.diff(st[:, :, 1], axis=1)
    print(wl, "shader clock MHz: median", np.median(cyc / real), "p10", np.percentile(cyc / real, 10), "p90", np.percentile(cyc / real, 90))
    eng.close()
