"""
The DEVICE truncated-normal sampler (csrc/d3d_rng.h, the Gibbs step's draw,
lib/run.py:495-496 -> lib/rtnorm.py:21-92) tested directly through d3d_rtnorm:
against the analytic distribution and against draws of the reference's own
rtnorm committed in tests/golden/ref_rtnorm.npz -- body, the alpha >= 6 tail
(Robert's translated exponential), the body/tail seam, mirrored intervals.
"""
import os

import numpy as np
import pytest
from scipy import stats

from deconv3d_amd import _lib
from oracle import deconv3d_oracle as O

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
N = 20000


@pytest.fixture(scope="module")
def eng():
    with _lib.Engine((4, 3, 3), (1, 1)) as e:
        yield e


def truncnorm_cdf(x, lo, hi, mu, sigma):
    a, b = (lo - mu) / sigma, (hi - mu) / sigma
    z = (np.asarray(x) - mu) / sigma
    if a + b > 0:
        # interval on the upper side of the mean: survival functions are well conditioned
        sa, sb, sx = stats.norm.sf(a), stats.norm.sf(b), stats.norm.sf(z)
        return (sa - sx) / (sa - sb)
    ca, cb, cx = stats.norm.cdf(a), stats.norm.cdf(b), stats.norm.cdf(z)
    return (cx - ca) / (cb - ca)


CASES = [  # (lo, hi, mu, sigma, label)
    (0.0, 50.0, 1.0, 1.0, "body"),
    (0.0, 9.0, 4.0, 3.0, "two-sided wide"),
    (-1.0, 2.0, 0.0, 1.0, "straddles the mode"),
    (0.0, 0.5, 2.0, 1.5, "narrow"),
    (0.0, 50.0, -2.9, 1.0, "lower bound at 2.9 sigma (erfc branch)"),
    (0.0, 50.0, -5.99, 1.0, "just below the tail seam"),
    (0.0, 50.0, -6.0, 1.0, "the tail seam, alpha = 6"),
    (0.0, 50.0, -8.0, 1.0, "far tail"),
    (0.0, 0.3, -7.0, 1.0, "tail with a near upper bound"),
    (0.0, 30.0, 45.0, 4.0, "mass against the upper bound"),
    (-50.0, 0.0, 8.0, 1.0, "mirrored far tail (beta <= 0)"),
    (-9.0, -1.0, 2.0, 1.0, "mirrored body"),
    (0.0, 12.5, 3.0, 1e-3, "tiny sigma"),
]


@pytest.mark.parametrize("lo,hi,mu,sigma,label", CASES)
def test_device_sampler_follows_the_truncated_normal(eng, lo, hi, mu, sigma, label):
    x = eng.rtnorm(lo, hi, mu, sigma, size=N, seed=2024)
    assert x.shape == (N,) and np.all(x >= lo) and np.all(x <= hi), label
    ks = stats.kstest(x, lambda t: truncnorm_cdf(t, lo, hi, mu, sigma))
    assert ks.pvalue > 1e-3, (label, ks)
    # the wavefront-cooperative form of the MH kernel draws the same numbers
    xw = eng.rtnorm(lo, hi, mu, sigma, size=2048, seed=2024, wave_mode=True)
    np.testing.assert_array_equal(xw, x[:2048])
    # and so does the oracle's restatement, draw by draw
    for i in (0, 1, 777):
        blocks = iter(range(2, 1000))

        def draw():
            return O.philox_pair(2024, i, 0, next(blocks))
        want = O.truncated_normal(lo, hi, mu, sigma, draw)
        assert abs(x[i] - want) <= 1e-9 * max(1.0, abs(want)), (label, i)


def test_device_sampler_against_the_reference_rtnorm_draws(eng):
    """tests/golden/ref_rtnorm.npz: 6000 sorted draws of the reference's rtnorm
    per regime (make_goldens.py).  Two-sample KS against the device.  The last
    regime is the one where the reference's python port is itself biased
    (DESIGN.md section 4): there the device must follow the analytic law and
    DIFFER from the reference."""
    g = np.load(os.path.join(GOLD, "ref_rtnorm.npz"))
    cases = g["cases"]
    for i, (a, b, mu, sg) in enumerate(cases):
        ref = g["draws_%d" % i]
        x = eng.rtnorm(a, b, mu, sg, size=6000, seed=31 + i)
        ks = stats.ks_2samp(x, ref)
        if i < len(cases) - 1:
            assert ks.pvalue > 1e-3, (i, tuple(cases[i]), ks)
        else:
            assert ks.pvalue < 1e-6
            assert stats.kstest(x, lambda t: truncnorm_cdf(t, a, b, mu, sg)).pvalue > 1e-3


def test_rtnorm_argument_checks(eng):
    with pytest.raises(ValueError):          # lib/rtnorm.py:66-68
        eng.rtnorm(1.0, 1.0, 0.0, 1.0, size=4)
    with pytest.raises(ValueError):
        eng.rtnorm(0.0, 1.0, 0.0, 0.0, size=4)
    assert eng.rtnorm(0.0, 1.0, 0.5, 1.0, size=0).shape == (0,)
