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
