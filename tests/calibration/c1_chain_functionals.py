"""Round-2 exploration: chain tolerance on well-identified functionals (C1)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from deconv3d_amd import _lib
from oracle import deconv3d_oracle as O

D, H, W = 32, 16, 16
fsf = O.gaussian_fsf_image(3.0); lsf = O.gaussian_lsf_vector(D, 0.9088)
data, var, mask, truth, init, mn, mx = O.synthetic_case(D, H, W, fsf, lsf, seed=12345)
ra = float(mx[0] ** 2)

def functionals(eng, ch):
    """per sample: convolved model at the 200 brightest data voxels, block fluxes"""
    idx = np.argsort(data.ravel())[-200:]
    out = np.empty((ch.shape[0], 200 + 16))
    for i, p in enumerate(ch):
        sim = eng.simulate(p, convolved=True)
        out[i, :200] = sim.ravel()[idx]
        flux = p[..., 0] * p[..., 2]
        out[i, 200:] = flux.reshape(4, 4, 4, 4).sum(axis=(1, 3)).ravel()
    return out

def bm(f, nb):
    n = (f.shape[0] // nb) * nb
    f = f[-n:]
    b = f.reshape(nb, -1, f.shape[1]).mean(1)
    return f.mean(0), b.std(0, ddof=1) / np.sqrt(nb)

with _lib.Engine((D, H, W), fsf.shape) as eng:
    eng.set_taps(fsf, lsf); eng.set_data(data, var, mask=mask); eng.set_params(init)
    eng.mh_config(mn, mx, 0.1, ra, seed=1, refresh_every=1000)
    eng.mh_sweeps(6000, 1)                       # burn in
    start = eng.get_params()
    res = {}
    for seed, n in ((11, 3000), (22, 3000), (33, 600)):
        eng.set_params(start); eng.mh_config(mn, mx, 0.1, ra, seed=seed, refresh_every=1000)
        ch = np.full((n + 1, H, W, 3), np.nan)
        eng.mh_sweeps(n, 1, 1, ch, None)
        res[seed] = functionals(eng, ch[1:])
    t0 = time.time()
    st = O.MHState(data, var, mask, fsf, lsf, start, mn, mx, seed=777)
    NO = 600
    och = np.empty((NO, H, W, 3))
    for s in range(1, NO + 1):
        O.mh_sweep(st, s); och[s - 1] = st.params
    print("oracle %d sweeps %.1fs" % (NO, time.time() - t0))
    res["oracle"] = functionals(eng, och)
for a, b, nb_a, nb_b in ((11, 22, 20, 20), (11, 33, 20, 10), (11, "oracle", 20, 10), (22, "oracle", 20, 10)):
    ma, sa = bm(res[a], nb_a); mb, sb = bm(res[b], nb_b)
    z = (ma - mb) / np.sqrt(sa ** 2 + sb ** 2)
    print("%s vs %s: model voxels rms z %.2f frac<3 %.3f max %.1f | block flux rms z %.2f max %.1f | rel diff model %.2e"
          % (a, b, np.sqrt(np.mean(z[:200] ** 2)), np.mean(np.abs(z[:200]) < 3), np.abs(z[:200]).max(),
             np.sqrt(np.mean(z[200:] ** 2)), np.abs(z[200:]).max(),
             np.abs(ma[:200] - mb[:200]).max() / np.abs(ma[:200]).max()))
