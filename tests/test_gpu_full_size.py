"""
Full-size (BASELINE config 3: 300x300x128, Moffat 11x11, 17-tap LSF) GPU tests:
size-independent properties (linearity and flux conservation of the separable
convolution, agreement of its two device paths, consistency of the carried residual
with a from-scratch one, lib/run.py:521-534, reproducibility of the chain under a
seed, a sub-region of the forward model against the oracle) -- and, in
tests/test_gpu_full_size_oracle.py, whole sweeps update by update against the oracle
(about 20 s of oracle per 90 000-update sweep).
"""
import numpy as np
import pytest

from deconv3d_amd import _lib
from deconv3d_amd.spread_functions import moffat_image, muse_like_lsf_vector
from oracle import deconv3d_oracle as O

pytestmark = pytest.mark.gpu

D, H, W = 128, 300, 300


@pytest.fixture(scope="module")
def problem():
    fsf = moffat_image((11, 11), beta=2.5, fwhm_px=3.0)
    lsf = muse_like_lsf_vector(D, sigma_px=0.9, box_px=1.0)
    rng = np.random.default_rng(12345)
    y, x = np.indices((H, W))
    truth = np.dstack((10.0 * np.exp(-((y - H / 2.) ** 2 + (x - W / 2.) ** 2) / (2. * (H / 6.) ** 2)),
                       D / 2. + (D / 8.) * np.tanh((x - W / 2.) / (W / 8.)),
                       rng.uniform(1.5, 3.0, size=(H, W))))
    eng = _lib.Engine((D, H, W), fsf.shape)
    eng.set_taps(fsf, lsf)
    eng.set_params(truth)
    clean = eng.forward()
    sigma = 0.5 * fsf.max()
    data = clean + rng.normal(0., sigma, size=clean.shape)
    var = np.full(clean.shape, sigma ** 2)
    max_b = np.array([data.max() / fsf.max(), D - 1., float(D)])
    init = max_b * rng.random((H, W, 3))
    eng.set_data(data, var)
    yield dict(eng=eng, fsf=fsf, lsf=lsf, truth=truth, data=data, var=var, init=init,
               min_b=np.zeros(3), max_b=max_b, rng=rng, clean=clean)
    eng.close()


def test_convolution_is_linear_conserves_flux_and_paths_agree(problem):
    eng, rng = problem["eng"], problem["rng"]
    a = rng.normal(size=(D, H, W))
    b = rng.normal(size=(D, H, W))
    ca, cb, cab = eng.convolve(a), eng.convolve(b), eng.convolve(2.0 * a - 3.0 * b)
    scale = np.max(np.abs(cab))
    assert np.max(np.abs(cab - (2.0 * ca - 3.0 * cb))) <= 1e-12 * scale
    # a point source well inside the cube keeps its flux (taps sum to 1)
    pt = np.zeros((D, H, W))
    pt[60, 150, 150] = 7.0
    cp = eng.convolve(pt)
    assert abs(cp.sum() - 7.0) <= 1e-12 * 7.0
    np.testing.assert_allclose(cp[:, 145:156, 145:156].sum(0), 7.0 * problem["fsf"][::-1, ::-1],
                               rtol=1e-12, atol=1e-15)
    # reference-layout path (d3d_convolve) == slot path (d3d_convolve_slots)
    eng.upload_slot(_lib.SLOT_TMP0, a)
    eng.convolve_slots(_lib.SLOT_TMP0, _lib.SLOT_SIM)
    assert np.max(np.abs(eng.download_slot(_lib.SLOT_SIM) - ca)) <= 1e-12 * np.max(np.abs(ca))


def test_forward_model_subregion_matches_oracle(problem):
    """A 24x26 patch of the full-size forward model against the oracle run on the
    patch plus its FSF margin."""
    y0, y1, x0, x1, m = 100, 124, 40, 66, 5
    sub = problem["truth"][y0 - m:y1 + m, x0 - m:x1 + m]
    ref = O.forward_full((D,) + sub.shape[:2], sub, np.ones(sub.shape[:2]), problem["fsf"],
                         problem["lsf"])[:, m:-m, m:-m]
    got = problem["clean"][:, y0:y1, x0:x1]
    assert np.max(np.abs(got - ref)) <= 1e-12 * np.max(np.abs(ref))


def test_chain_reproducible_and_residual_consistent(problem):
    eng = problem["eng"]
    outs = []
    for _ in range(2):
        eng.set_params(problem["init"])
        eng.mh_config(problem["min_b"], problem["max_b"], 0.1, float(problem["max_b"][0] ** 2),
                      seed=777, refresh_every=0)
        acc = eng.mh_sweeps(6, 1)
        outs.append((acc, eng.get_params()))
    assert outs[0][0] == outs[1][0] > 0
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    assert np.isfinite(outs[0][1]).all()
    assert (outs[0][1][..., 0] >= 0).all() and (outs[0][1][..., 0] <= problem["max_b"][0]).all()
    carried = eng.download_slot(_lib.SLOT_ERR)          # flushes the pending colour
    fresh = eng.residual()
    assert np.max(np.abs(carried - fresh)) <= 1e-11 * np.max(np.abs(fresh))
    # chi2 map of the carried residual == host evaluation
    cmap, total = eng.chi2_map()
    ref = 0.5 * np.sum(fresh ** 2 / problem["var"], axis=0)
    np.testing.assert_allclose(cmap, ref, rtol=1e-10)
    np.testing.assert_allclose(total, ref.sum(), rtol=1e-10)


@pytest.mark.parametrize("uniform", [False, True])
def test_pending_layer_depths_are_bit_identical_at_full_size(problem, uniform):
    """k_mh_ws (a kernel boundary after every colour) with the residual written back every
    colour, every second and every third one -- and, in a `make EXPERIMENTS=1` build,
    k_mh_flow (one launch per sweep, sc1 hand-off across the XCDs) and k_mh_pair (two colour
    classes per launch): 270 000 updates, chains, residuals and delta maps bit-identical.
    A stale line anywhere in a window would show up here."""
    var = problem["var"] if uniform else problem["var"] * (
        0.75 + 0.5 * np.random.default_rng(5).random(problem["var"].shape))
    outs = []
    variants = [{"mh_layers": 1}, {"mh_layers": 2}, {"mh_layers": 3}]
    if _lib.has_experiments():
        variants += [{"mh_flow": 1}, {"mh_layers": 2, "mh_pair": 1}]
    for opts in variants:
        with _lib.Engine((D, H, W), problem["fsf"].shape, options=opts) as eng:
            eng.set_taps(problem["fsf"], problem["lsf"])
            eng.set_data(problem["data"], var)
            assert eng.variance_is_uniform() == uniform
            eng.set_params(problem["init"])
            eng.mh_config(problem["min_b"], problem["max_b"], 0.1,
                          float(problem["max_b"][0] ** 2), seed=4242, refresh_every=0)
            acc = eng.mh_sweeps(3, 1)
            outs.append((np.int64(acc), eng.get_params(), eng.get_dlog(),
                         eng.download_slot(_lib.SLOT_ERR)))
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            np.testing.assert_array_equal(a, b)


def test_window_probe_matches_oracle_on_full_cube(problem):
    eng, rng = problem["eng"], problem["rng"]
    eng.set_params(problem["init"])
    err = eng.residual()
    for (y, x) in [(0, 0), (299, 299), (150, 7), (3, 296), (151, 149)]:
        p_old = problem["init"][y, x]
        p_new = p_old + np.array([0., 0.6, 0.2])
        got = eng.window_stats(y, x, p_new)
        ref = O.window_stats(err, problem["var"], p_old, p_new, y, x, problem["fsf"], problem["lsf"])
        np.testing.assert_allclose(got[:3], ref[:3], rtol=1e-10, atol=1e-12 * max(ref[0], ref[1]))
        np.testing.assert_allclose(got[3:], ref[3:], rtol=1e-10,
                                   atol=1e-12 * max(abs(ref[3]), abs(ref[4])))
