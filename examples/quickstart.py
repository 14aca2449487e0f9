#!/usr/bin/env python
# coding=utf-8
"""
Quick start: deconvolve a synthetic MUSE-like cube on one MI355X.

    python examples/quickstart.py [iterations] [chains]

Builds a 64x64x64 cube of one Gaussian emission line per spaxel (a rotating
disc seen through the default MUSE instrument: Gaussian FSF, Gaussian LSF),
adds noise, runs the MH-within-Gibbs chain through the same `Run` call a user
of irap-omp/deconv3d would write (lib/run.py:95-109), and prints how well the
posterior means recover the line centres and widths.  With a FITS file:

    cube = Cube.from_fits('my_cube.fits')
    run = Run(cube, MUSE(fsf_fwhm=0.8841), mask=above_percentile(cube, 60), max_iterations=40000)
    run.save('my_run')
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deconv3d_amd import MUSE, Run, _lib  # noqa: E402

iterations = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
D = H = W = 64
rng = np.random.default_rng(7)
inst = MUSE()                                       # Gaussian FSF (1" seeing) + Gaussian LSF
y, x = np.indices((H, W))
r2 = (y - H / 2.) ** 2 + (x - W / 2.) ** 2
truth = np.dstack((12.0 * np.exp(-r2 / (2. * (H / 5.) ** 2)),             # amplitude
                   D / 2. + (D / 7.) * np.tanh((x - W / 2.) / (W / 7.)),   # centre: rotation curve
                   1.6 + 0.8 * np.exp(-r2 / (2. * (H / 8.) ** 2))))        # width

# the noiseless observation comes from the device forward model
probe = inst.build_cube(np.zeros((D, H, W)))
fsf, lsf = inst.fsf.as_image(probe), inst.lsf.as_vector(probe)
with _lib.Engine((D, H, W), fsf.shape) as eng:
    eng.set_taps(fsf, lsf)
    eng.set_params(truth)
    clean = eng.forward()
sigma = 0.02 * clean.max()
cube = inst.build_cube(clean + rng.normal(0., sigma, clean.shape))

t0 = time.perf_counter()
run = Run(cube, inst, variance=np.full(clean.shape, sigma ** 2), max_iterations=iterations,
          keep_one_in=10, min_acceptance_rate=0., jump_amplitude=[0., 0.5, 0.2],
          gibbs_apriori_variance=100., seed=1)   # amplitude prior, as tests/read_mat.py:109-119 sets one
dt = time.perf_counter() - t0
burn = run.chain.shape[0] // 2
post = run.chain[burn:].mean(axis=0)                # posterior means over the second half
last = run.simulate_convolved(cube.data.shape, run.chain[-1])      # forward model of the last sample
model = run.convolved_cube.data                     # ... and of the extracted (mean) parameters
print("%d iterations of %d spaxels in %.1f s (%.2f M spaxel-updates/s including setup)" % (
    iterations, H * W, dt, iterations * H * W / dt / 1e6))
print("reduced chi2: last sample %.4f, extracted parameters %.4f; their convolved model vs the noiseless cube: "
      "rms %.2f %% of its peak" % (np.mean(((cube.data - last) / sigma) ** 2),
                                   np.mean(((cube.data - model) / sigma) ** 2),
                                   100. * np.sqrt(np.mean((model - clean) ** 2)) / clean.max()))
# Per-spaxel parameters are what a 13x13-pixel seeing leaves of them: neighbours trade
# flux, so single spaxels scatter far more than the convolved model does.
bright = truth[..., 0] > 3.0
print("bright spaxels (%d): median |centre - truth| = %.2f channels, |width - truth| = %.2f channels" % (
    bright.sum(), np.median(np.abs(post[..., 1] - truth[..., 1])[bright]),
    np.median(np.abs(post[..., 2] - truth[..., 2])[bright])))


# ---- several chains at once: chains=R ----------------------------------------------------------
# The reference's own science fixture (tests/input/data14forAntoine.mat: 24 x 30 spaxels x 21
# channels, settings of its tests/read_mat.py:94-121).  A colour launch of so small a cube holds
# two or three windows -- a single chain is a chain of launch latencies -- so R independent chains
# (seeds seed + r) advance TOGETHER, one launch per colour class for all of them, in about the
# time of one; run.chains[r] is bit for bit the chain of Run(..., seed=seed + r), the posterior
# means pool the chains and run.rhat says whether they agree.
from deconv3d_amd import above_percentile  # noqa: E402

chains = int(sys.argv[2]) if len(sys.argv) > 2 else 8
fixture = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                       "ref_mat_fixture.npz")
g = np.load(fixture)
inst2 = MUSE(fsf_fwhm=0.8841, lsf_fwhm=0.)
cube2 = inst2.build_cube(g["data"])
kw = dict(variance=g["var"], gibbs_apriori_variance=5., mask=above_percentile(cube2, 60),
          max_iterations=min(iterations, 4000), keep_one_in=10, seed=7, min_acceptance_rate=0.)
one = Run(cube2, inst2, **kw)
many = Run(cube2, inst2, chains=chains, **kw)
live = many.mask == 1
print("reference fixture %s: 1 chain %.2f s of sweeps, %d chains %.2f s (%.1fx the samples per second); "
      "chain 0 identical: %s; R-hat of the live spaxels' (a, c, w): median %s, 95th percentile %s" % (
          "x".join(str(v) for v in g["data"].shape), one.mh_seconds, chains, many.mh_seconds,
          chains * one.mh_seconds / many.mh_seconds, bool(np.array_equal(one.chain, many.chains[0])),
          np.round(np.nanmedian(many.rhat[live], axis=0), 3), np.round(np.nanpercentile(many.rhat[live], 95, axis=0), 3)))
