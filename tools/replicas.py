"""Aggregate rate of R independent chains of a small cube on one GPU (profiles/r03_replicas.txt).
    python tools/replicas.py [workload]"""
import os, sys, time, threading
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import bench as B
from deconv3d_amd import _lib
wl = sys.argv[1] if len(sys.argv) > 1 else "c2_64x64x64"
D, H, W, fs = B.WORKLOADS[wl]
fsf, lsf = B.build_taps(D, fs)
def make(seed):
    eng = _lib.Engine((D, H, W), fsf.shape)
    eng.set_taps(fsf, lsf)
    data, var, truth, init, min_b, max_b = B.synthetic_inputs(eng, D, H, W, fsf, seed)
    eng.set_data(data, var)
    eng.set_params(init)
    eng.mh_config(min_b, max_b, 0.1, float(max_b[0] ** 2), seed=seed, refresh_every=0)
    eng.residual(fetch=False)
    eng.mh_sweeps(10, 1)
    return eng
nsw = 300
for R in (1, 2, 4, 8, 16):
    engs = [make(100 + r) for r in range(R)]
    def work(e):
        e.mh_sweeps(nsw, 11)
    for e in engs: e.sync()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(e,)) for e in engs]
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print("%s: %2d chains on one GPU (one host thread each): %.3f ms per sweep of all chains, %.2f M updates/s in total" % (wl, R, dt * 1e3 / nsw, R * nsw * H * W / dt / 1e6), flush=True)
    # phase-interleaved from ONE host thread
    for e in engs: e.sync()
    t0 = time.perf_counter()
    for s in range(nsw):
        for e in engs: e.mh_phase(0, 400 + s)
    for e in engs: e.sync()
    dt = time.perf_counter() - t0
    print("%s: %2d chains, one host thread interleaving sweeps: %.3f ms per sweep of all chains, %.2f M updates/s in total" % (wl, R, dt * 1e3 / nsw, R * nsw * H * W / dt / 1e6), flush=True)
    for e in engs: e.close()

# batched: R chains of one geometry in ONE launch per colour class (d3d_mh_sweeps_batch)
for R in (1, 4, 8, 16, 32, 64):
    if R * D * H * W * 8 * 10 > 40e9:
        break
    engs = [make(100 + r) for r in range(R)]
    _lib.mh_sweeps_batch(engs, 10, 11)
    t0 = time.perf_counter()
    _lib.mh_sweeps_batch(engs, nsw, 21)
    dt = time.perf_counter() - t0
    print("%s: %2d chains batched into one launch per colour class: %.3f ms per sweep of all chains, %.2f M updates/s in total" % (wl, R, dt * 1e3 / nsw, R * nsw * H * W / dt / 1e6), flush=True)
    for e in engs: e.close()
