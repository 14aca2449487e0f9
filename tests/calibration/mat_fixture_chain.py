"""Round-2 exploration on the GPU box: statistics that calibrate the science tests."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import deconv3d_amd as d3d
from deconv3d_amd import _lib
from oracle import deconv3d_oracle as O

g = np.load(os.path.join(ROOT, "tests/golden/ref_mat_fixture.npz"))
data, var, fsf15, truth = g["data"], g["var"], g["fsf"], g["params"]
D, H, W = data.shape
print("mat fixture", data.shape, "a range", truth[..., 0].min(), truth[..., 0].max())
inst_ref = d3d.MUSE(fsf_fwhm=0.8841, lsf_fwhm=0.)
cube = inst_ref.build_cube(data)
mask = d3d.above_percentile(cube, 60)
print("mask live", int(mask.sum()), "fsf", inst_ref.fsf.as_image(cube).shape)
for label, inst in (("read_mat.py instrument (Gaussian fwhm 0.8841\")", inst_ref),
                    ("the fixture's own 15x15 FSF", d3d.Instrument(
                        lsf=d3d.VectorLineSpreadFunction(inst_ref.lsf.as_vector(cube)),
                        fsf=d3d.ImageFieldSpreadFunction(fsf15)))):
    for n in (2000, 8000):
        t0 = time.time()
        run = d3d.Run(cube, inst, variance=var, gibbs_apriori_variance=5., mask=mask,
                      max_iterations=n, keep_one_in=10, seed=7, min_acceptance_rate=0.)
        dt = time.time() - t0
        live = mask == 1
        tail = run.chain[int(0.8 * run.chain.shape[0]):]
        mean, std = tail.mean(0), tail.std(0)
        run.engine.set_params(run.chain[-1]); err = run.engine.residual()
        win = np.zeros((H, W), bool)
        red_all = np.sum(err ** 2 / var) / err.size
        red_live = np.sum((err ** 2 / var)[:, live]) / (D * live.sum())
        bright = live & (truth[..., 0] > np.percentile(truth[..., 0][live], 50))
        dev = (mean - truth)
        z = dev / np.maximum(std, 1e-12)
        print("%s | %d sweeps %.1fs acc %.3f | red chi2 all %.4f live %.4f | bright n=%d "
              "median |da| %.3f |dc| %.3f |dw| %.3f | rms z a %.2f c %.2f w %.2f | frac|z|<3 %.3f"
              % (label, n, dt, run.acceptance_rate, red_all, red_live, bright.sum(),
                 np.median(np.abs(dev[..., 0][bright])), np.median(np.abs(dev[..., 1][bright])),
                 np.median(np.abs(dev[..., 2][bright])),
                 np.sqrt(np.mean(z[..., 0][bright] ** 2)), np.sqrt(np.mean(z[..., 1][bright] ** 2)),
                 np.sqrt(np.mean(z[..., 2][bright] ** 2)), np.mean(np.abs(z[bright]) < 3)))
        sys.stdout.flush()

# ---- C1 chain tolerance: device vs oracle, different seeds -------------------
D, H, W = 32, 16, 16
fsf = O.gaussian_fsf_image(3.0); lsf = O.gaussian_lsf_vector(D, 0.9088)
data, var, mask, truth, init, mn, mx = O.synthetic_case(D, H, W, fsf, lsf, seed=12345)
N = 400
t0 = time.time()
st = O.MHState(data, var, mask, fsf, lsf, init, mn, mx, seed=777)
och = np.empty((N + 1, H, W, 3)); och[0] = init
for s in range(1, N + 1):
    O.mh_sweep(st, s); och[s] = st.params
print("oracle %d sweeps %.1fs" % (N, time.time() - t0))
with _lib.Engine((D, H, W), fsf.shape) as eng:
    eng.set_taps(fsf, lsf); eng.set_data(data, var, mask=mask); eng.set_params(init)
    eng.mh_config(mn, mx, 0.1, st.ra, seed=4242, refresh_every=0)
    dch = np.full((N + 1, H, W, 3), np.nan); dch[0] = init
    eng.mh_sweeps(N, 1, 1, dch, None)
def bm(ch, nb=10):
    m = ch.mean(0); b = ch.reshape(nb, -1, *ch.shape[1:]).mean(1)
    return m, b.std(0, ddof=1) / np.sqrt(nb)
for burn in (100, 200):
    mo, so = bm(och[burn + 1:]); md, sd = bm(dch[burn + 1:])
    z = (md - mo) / np.sqrt(so ** 2 + sd ** 2)
    for k, nm in enumerate("acw"):
        print("C1 burn %d param %s: rms z %.2f frac|z|<3 %.3f max|z| %.1f" %
              (burn, nm, np.sqrt(np.mean(z[..., k] ** 2)), np.mean(np.abs(z[..., k]) < 3),
               np.abs(z[..., k]).max()))
