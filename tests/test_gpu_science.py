"""
Science-level checks of the device chain.

  * the reference's own known-answer data set (tests/input/data14forAntoine.mat +
    Parametres_theoriques.mat, committed as tests/golden/ref_mat_fixture.npz) run
    with the settings of the reference's tests/read_mat.py:94-121 -- the numerical
    form of tests/analyze_run.py:16-54 (chain vs theoretical parameters);
  * SURVEY 8(d) chain tolerance: posterior means of the device chain against the
    CPU oracle's chain at BASELINE config 1, different seeds, within 3 MC sigma;
  * BASELINE config 2 at exactly 64x64x64 / Moffat 11x11 against the oracle;
  * the device LSF pass against the reference-saved FITS pair.
"""
import os

import numpy as np
import pytest

import deconv3d_amd as d3d
from deconv3d_amd import _lib
from oracle import deconv3d_oracle as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


def test_device_lsf_pass_against_the_reference_saved_pair():
    """The z factor of the older revision's separable 3-D convolution is
    convolve_1d (tests/test_oracle.py).  So: DEVICE LSF pass (depth 30: the
    partial-wrap branch of lib/convolution.py:89-160) on the reference's saved
    deconvolved cube, then the legacy spatial factor on the host, must give the
    reference's saved convolved cube."""
    g = gold("ref_galpak_pair.npz")
    clean, conv = g["clean"], g["convolved"]
    D, H, W = clean.shape
    lsf = O.gaussian_lsf_vector(D, 2.675 / 2.35482 / 1.25)
    with _lib.Engine((D, H, W), (1, 1)) as eng:
        eng.set_taps(np.ones((1, 1)), lsf)
        tmp = eng.convolve(clean)
    fsf_full = O.legacy_gaussian_fsf_full((H, W), 1.0 / 0.2)
    out = np.stack([O.legacy_convolve_2d_same(tmp[z], fsf_full) for z in range(D)])
    assert np.abs(out - conv).max() <= 1e-12 * conv.max()


def test_config2_exact_shape_against_the_oracle():
    """BASELINE config 2 at its full footprint: 64x64x64, Moffat beta 2.5 FWHM 3 px
    cropped 11x11, Gaussian LSF.  Forward model, two sweeps of the
    chain (8192 updates), log acceptance ratios, accepted count and the carried
    residual against the CPU oracle."""
    D, H, W = 64, 64, 64
    fsf = O.moffat_cropped(11, 3.0, 2.5)
    lsf = O.gaussian_lsf_vector(D, 0.9088)
    data, var, mask, truth, init, mn, mx = O.synthetic_case(D, H, W, fsf, lsf, seed=12345)
    # heteroscedastic variance: the general kernel (the uniform variant is covered elsewhere)
    rng = np.random.default_rng(5)
    var = var * rng.uniform(0.75, 1.25, size=var.shape)
    st = O.MHState(data, var, mask, fsf, lsf, init, mn, mx, seed=2024)
    with _lib.Engine((D, H, W), fsf.shape) as eng:
        eng.set_taps(fsf, lsf)
        eng.set_data(data, var, mask=mask)
        eng.set_params(truth)
        sim = eng.forward()
        want = O.forward_full((D, H, W), truth, mask, fsf, lsf)
        assert np.max(np.abs(sim - want)) <= 1e-12 * np.max(np.abs(want))
        eng.set_params(init)
        eng.mh_config(mn, mx, 0.1, st.ra, seed=2024, refresh_every=0)
        chain = np.full((3, H, W, 3), np.nan)
        dlog = np.full((3, H, W), np.nan)
        acc = eng.mh_sweeps(2, 1, 1, chain, dlog)
        for s in (1, 2):
            O.mh_sweep(st, s)
            np.testing.assert_allclose(chain[s], st.params, rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(dlog[s], st.dlog, rtol=1e-8,
                                       atol=1e-10 * (np.abs(st.dlog).max() + 1))
        assert acc == st.accepted
        err = eng.download_slot(_lib.SLOT_ERR)
        assert np.max(np.abs(err - st.err)) <= 1e-11 * np.max(np.abs(st.err))


def test_reference_mat_fixture_chain_recovers_the_theoretical_parameters():
    """The reference's own science check (tests/read_mat.py:94-121 runs the chain,
    tests/analyze_run.py:16-54 plots it against Parametres_theoriques.mat), made
    numerical: data14forAntoine.mat through Run() with read_mat.py's settings --
    MUSE(fsf_fwhm=0.8841, lsf_fwhm=0), the fixture's variance cube,
    mask = above_percentile(cube, 60), gibbs_apriori_variance = 5, keep_one_in = 10
    -- from the default RANDOM start (lib/run.py:310-314).  The posterior (last 20 %
    of the chain, as Run.extract_parameters) must contain the theoretical
    parameters, and the fit must explain the data: reduced chi2 ~ 1."""
    import logging
    logging.getLogger("deconv3d").setLevel(logging.WARNING)
    g = gold("ref_mat_fixture.npz")
    data, var, truth = g["data"], g["var"], g["params"]
    D, H, W = data.shape
    inst = d3d.MUSE(fsf_fwhm=0.8841, lsf_fwhm=0.)
    cube = inst.build_cube(data)
    mask = d3d.above_percentile(cube, 60)
    run = d3d.Run(cube, inst, variance=var, gibbs_apriori_variance=5., mask=mask,
                  max_iterations=8000, keep_one_in=10, seed=7, min_acceptance_rate=0.)
    assert run.fsf.shape == (13, 13) and run.chain.shape == (800, H, W, 3)
    live = mask == 1
    assert live.sum() == 288
    # reduced chi2 of the last state over the live spaxels' spectra
    run.engine.set_params(run.chain[-1])
    err = run.engine.residual()
    red = np.sum((err ** 2 / var)[:, live]) / (D * live.sum())
    assert 0.95 < red < 1.06, red
    # posterior of the last 20 % of the saved chain vs the theoretical parameters, at
    # the brighter half of the live spaxels (measured: rms z 0.7 / 0.4 / 0.5, 99.8 %)
    tail = run.chain[int(0.8 * run.chain.shape[0]):]
    mean, std = tail.mean(0), tail.std(0)
    np.testing.assert_allclose(run.parameters[live], mean[live], rtol=1e-12, atol=1e-12)
    bright = live & (truth[..., 0] > np.percentile(truth[..., 0][live], 50))
    z = (mean - truth)[bright] / std[bright]
    assert np.all(np.sqrt(np.mean(z ** 2, axis=0)) < 1.6), np.sqrt(np.mean(z ** 2, axis=0))
    assert np.mean(np.abs(z) < 3) > 0.98
    dev = np.abs(mean - truth)[bright]
    assert np.median(dev[:, 1]) < 0.4 and np.median(dev[:, 2]) < 0.35, np.median(dev, axis=0)
    # the 1-based centres of the Matlab file are clearly worse (tests/read_mat.py: c -= 1)
    assert np.median(np.abs(mean[..., 1] - (truth[..., 1] + 1.0))[bright]) > 0.6


def test_chain_tolerance_device_against_oracle_config1():
    """SURVEY 8(d) chain tolerance at BASELINE config 1 (32x16x16, Gaussian 9x9):
    posterior means of the DEVICE chain and of the CPU ORACLE chain, different
    seeds, agree within 3 Monte-Carlo sigma.  The posterior is ill conditioned
    (neighbouring amplitudes trade against each other under a 3-px FSF; chains
    mix over hundreds of sweeps), so MC sigma is not taken from batch means of one
    chain -- which underestimate it severalfold here -- but from REPLICATES: K
    device chains with different seeds from a common converged state, against
    which the one oracle chain (other seed, same start, same length) must be
    exchangeable.  z = (oracle mean - device mean) / (replicate sd * sqrt(1 + 1/K))
    follows Student's t with K-1 degrees of freedom, for (a, c, w) of every bright
    spaxel, the convolved model at the 200 brightest voxels and 4x4 block fluxes."""
    D, H, W = 32, 16, 16
    fsf, lsf = O.gaussian_fsf_image(3.0), O.gaussian_lsf_vector(D, 0.9088)
    data, var, mask, truth, init, mn, mx = O.synthetic_case(D, H, W, fsf, lsf, seed=12345)
    ra = float(mx[0] ** 2)
    K, N = 8, 500
    bright = truth[..., 0] > 3.0
    top = np.argsort(data.ravel())[-200:]

    def functionals(eng, chain):
        f = np.zeros(bright.sum() * 3 + 200 + 16)
        for p in chain:
            f[:bright.sum() * 3] += p[bright].ravel()
            f[bright.sum() * 3:-16] += eng.simulate(p, convolved=True).ravel()[top]
            f[-16:] += (p[..., 0] * p[..., 2]).reshape(4, 4, 4, 4).sum(axis=(1, 3)).ravel()
        return f / len(chain)

    with _lib.Engine((D, H, W), fsf.shape) as eng:
        eng.set_taps(fsf, lsf)
        eng.set_data(data, var, mask=mask)
        eng.set_params(init)
        eng.mh_config(mn, mx, 0.1, ra, seed=1, refresh_every=1000)
        eng.mh_sweeps(6000, 1)                                   # burn-in on the device
        start = eng.get_params()
        reps = []
        for k in range(K):
            eng.set_params(start)
            eng.mh_config(mn, mx, 0.1, ra, seed=100 + k, refresh_every=0)
            chain = np.full((N + 1, H, W, 3), np.nan)
            eng.mh_sweeps(N, 1, 1, chain, None)
            reps.append(functionals(eng, chain[1:]))
        st = O.MHState(data, var, mask, fsf, lsf, start, mn, mx, seed=777)
        och = np.empty((N, H, W, 3))
        for s in range(1, N + 1):
            O.mh_sweep(st, s)
            och[s - 1] = st.params
        orc = functionals(eng, och)
    reps = np.array(reps)
    z = (orc - reps.mean(0)) / (reps.std(0, ddof=1) * np.sqrt(1.0 + 1.0 / K))
    nb = int(bright.sum()) * 3
    groups = {"spaxel parameters": z[:nb], "model voxels": z[nb:-16], "block fluxes": z[-16:]}
    for name, zz in groups.items():
        print("chain tolerance, %s: rms z %.2f (t_%d expects %.2f), median |z| %.2f, |z| > 3: %.3f, "
              "max |z| %.1f" % (name, np.sqrt(np.mean(zz ** 2)), K - 1, np.sqrt((K - 1) / (K - 3.0)),
                                np.median(np.abs(zz)), np.mean(np.abs(zz) > 3), np.abs(zz).max()))
    # the well-identified functionals (what the data constrain): t_7 has rms 1.18 and
    # P(|z| > 3) = 0.02; they are correlated among themselves, hence the slack
    zz = np.concatenate((groups["model voxels"], groups["block fluxes"]))
    assert np.sqrt(np.mean(zz ** 2)) < 1.8
    assert np.mean(np.abs(zz) > 3) < 0.08
    # per-spaxel (a, c, w): heavy-tailed (a spaxel trades amplitude with its neighbours and
    # can sit in one mode for a whole chain), so quantiles rather than moments
    zp = np.abs(groups["spaxel parameters"])
    assert np.median(zp) < 1.3 and np.percentile(zp, 90) < 3.5, (np.median(zp), np.percentile(zp, 90))
    # and a leave-one-out control: a DEVICE replicate against the others behaves alike
    zc = (reps[0] - reps[1:].mean(0)) / (reps[1:].std(0, ddof=1) * np.sqrt(1.0 + 1.0 / (K - 1)))
    assert abs(np.median(np.abs(zc)) - np.median(np.abs(z))) < 0.5
