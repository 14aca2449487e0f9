"""
GPU tests of the host-evaluated LineModel path (custom python plugins,
lib/line_models.py:17-61): d3d_mh_colour_lines against a numpy restatement of
lib/run.py:391-519 for arbitrary unit lines, and Run() with custom models.
"""
import math

import numpy as np
import pytest

import deconv3d_amd as d3d
from deconv3d_amd import _lib
from oracle import deconv3d_oracle as O
from tests.cases import make_case

pytestmark = pytest.mark.gpu


def lorentz(x, c, g):
    return 1.0 / (1.0 + ((x - c) / g) ** 2)


@pytest.mark.parametrize("gibbs", [True, False])
def test_colour_lines_kernel_matches_numpy(gibbs):
    case = make_case("c1")
    D, H, W = case["D"], case["H"], case["W"]
    fsf, lsf, var = case["fsf"], case["lsf"], case["var"]
    fh, fw = fsf.shape
    rng = np.random.default_rng(3)
    x = np.arange(D, dtype=float)
    # a colour class: disjoint windows
    ys, xs = np.meshgrid(np.arange(2, H, fh), np.arange(3, W, fw), indexing="ij")
    ys, xs = ys.ravel(), xs.ravel()
    n = len(ys)
    a_old = 1.0 + 4 * rng.random(n) if gibbs else np.ones(n)
    lines = np.empty((n, 2, D))
    for i in range(n):
        lines[i, 0] = lorentz(x, D * (0.3 + 0.4 * rng.random()), 1 + 2 * rng.random())
        lines[i, 1] = lorentz(x, D * (0.3 + 0.4 * rng.random()), 1 + 2 * rng.random())
    in3 = np.column_stack((a_old, (rng.random(n) < 0.2).astype(float), np.log(rng.random(n))))
    err0 = rng.normal(size=(D, H, W))
    ra, seed, sweep = 30.0, 11, 4
    lo, hi = 0.0, 9.0
    with _lib.Engine((D, H, W), fsf.shape) as eng:
        eng.set_taps(fsf, lsf)
        eng.set_data(case["data"], var, mask=np.ones((H, W)))
        eng.mh_config([lo, 0, 0], [hi, 1, 1], [0, 0, 0], ra, seed=seed, refresh_every=0)
        eng.upload_slot(_lib.SLOT_ERR, err0)
        out = eng.mh_colour_lines(sweep, ys * W + xs, in3, lines, gibbs=gibbs)
        err1 = eng.download_slot(_lib.SLOT_ERR)
    ref_err = err0.copy()
    for i in range(n):
        y, xx = int(ys[i]), int(xs[i])
        (y0, y1, x0, x1), (ly0, ly1, lx0, lx1) = O.window_limits(y, xx, H, W, fh, fw)
        f = fsf[ly0:ly1, lx0:lx1]
        EO = O.spectral_convolve(lines[i, 0], lsf)
        EN = O.spectral_convolve(lines[i, 1], lsf)
        e = ref_err[:, y0:y1, x0:x1]
        v = var[:, y0:y1, x0:x1]
        c_old = a_old[i] * EO[:, None, None] * f
        c_new = a_old[i] * EN[:, None, None] * f
        ul = e + c_old
        delta = 0.5 * np.sum(e ** 2 / v) - 0.5 * np.sum((ul - c_new) ** 2 / v)
        accept = (in3[i, 2] < delta) and in3[i, 1] == 0.0
        E = EN if accept else EO
        ek = E[:, None, None] * f
        if gibbs:
            ro, mu, _, _ = O.gibbs_moments(ek, ul, v, ra)
            blk = [O.BLK_GIBBS]

            def draw():
                pair = O.philox_pair(seed, y * W + xx, sweep, blk[0])
                blk[0] += 1
                return pair

            r = O.truncated_normal(lo, hi, mu, math.sqrt(ro), draw)
        else:
            r = 1.0
        ref_err[:, y0:y1, x0:x1] = ul - ek * r
        assert bool(out[i, 0]) == accept
        np.testing.assert_allclose(out[i, 1], r, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(out[i, 2], delta, rtol=1e-9, atol=1e-9 * np.sum(e ** 2 / v))
    assert np.max(np.abs(err1 - ref_err)) <= 1e-11 * np.max(np.abs(ref_err))


class LorentzLineModel(d3d.LineModel):
    """a / (1 + ((x-c)/g)^2): a custom 3-parameter plugin."""

    def parameters(self):
        return ['a', 'c', 'g']

    def gibbs_parameter_index(self):
        return 0

    def min_boundaries(self, runner):
        return [0, 0, 0.3]

    def max_boundaries(self, runner):
        d = runner.cube.data
        return [np.amax(d) / np.amax(runner.fsf), d.shape[0] - 1, d.shape[0] / 4.]

    def modelize(self, runner, x, p):
        return p[0] * lorentz(np.asarray(x, float), p[1], p[2])


class TwoPeakModel(d3d.LineModel):
    """a * (gauss(c, w) + gauss(c + sep, w)): four parameters, amplitude Gibbs-sampled,
    and a post_jump hook that keeps the separation positive."""

    def parameters(self):
        return ['c', 'a', 'w', 'sep']

    def gibbs_parameter_index(self):
        return 1

    def min_boundaries(self, runner):
        return [0, 0, 0.5, 0]

    def max_boundaries(self, runner):
        d = runner.cube.data
        return [d.shape[0] - 1, np.amax(d) / np.amax(runner.fsf), 6, 10]

    def post_jump(self, runner, old, new):
        new[3] = abs(new[3])

    def modelize(self, runner, x, p):
        x = np.asarray(x, float)
        g = lambda c: np.exp(-(x - c) ** 2 / (2 * p[2] ** 2))
        return p[1] * (g(p[0]) + g(p[0] + p[3]))


def _cube_from(model, truth, inst, D, H, W, seed):
    rng = np.random.default_rng(seed)
    blank = inst.build_cube(np.ones((D, H, W)))
    fsf, lsf = inst.fsf.as_image(blank), inst.lsf.as_vector(blank)
    clean = np.zeros((D, H, W))
    for y in range(H):
        for x in range(W):
            clean[:, y, x] = model.modelize(None, np.arange(D), truth[y, x])
    with _lib.Engine((D, H, W), fsf.shape) as eng:
        eng.set_taps(fsf, lsf)
        conv = eng.convolve(clean)
    sigma = 0.03 * conv.max()
    return inst.build_cube(conv + rng.normal(0, sigma, conv.shape)), np.full(conv.shape, sigma ** 2)


@pytest.mark.parametrize("model_cls", [LorentzLineModel, TwoPeakModel])
def test_run_with_custom_line_model(model_cls):
    D, H, W = 32, 8, 8
    inst = d3d.MUSE(fsf_fwhm=0.5)
    rng = np.random.default_rng(1)
    if model_cls is LorentzLineModel:
        truth = np.dstack((3 + 3 * rng.random((H, W)), 10 + 10 * rng.random((H, W)),
                           1 + rng.random((H, W))))
    else:
        truth = np.dstack((8 + 8 * rng.random((H, W)), 3 + 3 * rng.random((H, W)),
                           1 + rng.random((H, W)), 4 + 2 * rng.random((H, W))))
    model = model_cls()
    cube, var = _cube_from(model, truth, inst, D, H, W, seed=2)
    run = d3d.Run(cube, inst, variance=var, model=model_cls, max_iterations=40, keep_one_in=2,
                  jump_amplitude=0.3, seed=5, initial_parameters=truth * 1.15)
    P = len(model.parameters())
    assert run.chain.shape == (20, H, W, P) and not np.isnan(run.chain).any()
    assert run.parameters.shape == (H, W, P)
    g = model.gibbs_parameter_index()
    assert (run.chain[..., g] >= 0).all()
    if model_cls is TwoPeakModel:
        assert (run.chain[1:, ..., 3] >= 0).all()              # post_jump hook honoured
    # the fit explains the data
    resid = cube.data - run.simulate_convolved(cube.data.shape, run.chain[-1])
    assert np.sum(resid ** 2 / var) / resid.size < 3.0
    start = cube.data - run.simulate_convolved(cube.data.shape, truth * 1.15)
    assert np.sum(resid ** 2) < 0.5 * np.sum(start ** 2)
    assert run.clean_cube.data.shape == cube.data.shape
    assert 0 < run.acceptance_rate <= 1


def test_custom_line_model_with_several_chains():
    """chains=R with a host-evaluated model: chain r is the chain of Run(..., seed=seed + r)."""
    D, H, W = 32, 8, 8
    inst = d3d.MUSE(fsf_fwhm=0.5)
    rng = np.random.default_rng(1)
    truth = np.dstack((3 + 3 * rng.random((H, W)), 10 + 10 * rng.random((H, W)), 1 + rng.random((H, W))))
    cube, var = _cube_from(LorentzLineModel(), truth, inst, D, H, W, seed=2)
    kw = dict(variance=var, model=LorentzLineModel, max_iterations=9, keep_one_in=2, jump_amplitude=0.3,
              initial_parameters=truth * 1.15)
    multi = d3d.Run(cube, inst, seed=5, chains=2, **kw)
    assert multi._host_model and multi.rhat.shape == (H, W, 3)
    for r in range(2):
        one = d3d.Run(cube, inst, seed=5 + r, **kw)
        np.testing.assert_array_equal(multi.chains[r], one.chain)
        np.testing.assert_array_equal(multi.all_likelihoods[r][1:], one.likelihoods[1:])
        assert multi.acceptance_rates[r] == one.acceptance_rate
    assert not np.array_equal(multi.chains[0][-1], multi.chains[1][-1])


def test_gaussian_subclass_with_own_bounds_stays_on_device():
    class Narrow(d3d.SingleGaussianLineModel):
        def max_boundaries(self, runner):
            b = d3d.SingleGaussianLineModel.max_boundaries(self, runner)
            b[2] = 4.0
            return b

    D, H, W = 16, 6, 6
    inst = d3d.MUSE(fsf_fwhm=0.5)
    cube = inst.build_cube(np.random.default_rng(0).random((D, H, W)) + 1)
    run = d3d.Run(cube, inst, model=Narrow, variance=np.full((D, H, W), 0.1), max_iterations=10)
    assert not run._host_model
    assert (run.chain[..., 2] <= 4.0 + 1e-12).all()
