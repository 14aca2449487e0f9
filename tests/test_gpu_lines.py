"""
k_lines_dense (csrc/d3d_kernels.h): the line cube -- lib/line_models.py:92-109, with the LSF of
lib/run.py:1011-1024 / lib/convolution.py:89-160 -- built by one lane group of a wavefront per
spaxel with dense LSF taps, against the oracle and against the tap-list kernel it replaces
(option lines_dense = 0): bit for bit with the library's exp (lines_dense = 1, the default where
the line kernel applies the LSF), within rounding with its own exp (2).  Depths chosen for every case of the padded grid: powers of
two (circular wrap), depths within 8 channels of one (partial wrap), ragged and odd depths,
several 128-channel steps, and spectra that share a wavefront.
"""
import numpy as np
import pytest

from deconv3d_amd import _lib
from oracle import deconv3d_oracle as O

pytestmark = pytest.mark.gpu

DELTA = np.zeros((3, 3))
DELTA[1, 1] = 1.0          # FSF pass = identity (products with zero taps add exact zeros)


def lsf_asym(D, rng):
    v = np.zeros(D)
    zc = (D - 1) // 2 - (D % 2 - 1)
    lo, hi = max(zc - 8, 0), min(zc + 9, D)
    v[lo:hi] = rng.random(hi - lo)
    return v / v.sum()


def line_cubes(D, H, W, lsf, params, mask, dense, rounds=0):
    """(clean, LSF-convolved) line cubes with lines_dense = `dense`; the LSF is applied by the
    line kernel itself (no fused epilogue in the FSF pass).  rounds: spaxel rounds per wavefront
    (0: by the cube's size -- one for cubes this small)."""
    if dense == 1:                             # (1 keeps the tap-list kernel for clean lines: the
        dense = 3                              #  test-only value 3 = k_lines_dense everywhere, library exp)
    opts = {"lines_dense": dense, "lines_rounds": rounds, "conv_rows": 0, "sep_fuse": 0}
    with _lib.Engine((D, H, W), DELTA.shape, options=opts) as eng:
        eng.set_taps(DELTA, lsf)
        eng.set_data(np.zeros((D, H, W)), np.ones((D, H, W)), mask=mask)
        return eng.simulate(params, convolved=False), eng.simulate(params, convolved=True)


@pytest.mark.parametrize("D,lsf_kind", [
    (128, "muse"), (128, "asym"), (128, "none"), (64, "asym"), (32, "muse"), (32, "asym"),
    (21, "muse"), (30, "asym"), (100, "asym"), (48, "muse"),
    (127, "asym"), (125, "asym"), (121, "asym"), (63, "asym"),       # partial wrap of the padded grid
    (256, "asym"), (255, "asym"), (249, "asym"), (200, "muse"), (130, "asym"), (301, "asym"),
    (384, "muse"), (1020, "asym"), (1024, "asym"),
])
def test_dense_line_kernel_matches_the_oracle_and_the_tap_list_kernel(D, lsf_kind):
    H, W = 7, 9
    rng = np.random.default_rng(D * 7 + len(lsf_kind))
    lsf = {"muse": O.muse_like_lsf, "asym": lambda d: lsf_asym(d, rng), "none": lambda d: None}[lsf_kind](D)
    params = np.dstack((1 + 9 * rng.random((H, W)), D * (-0.05 + 1.1 * rng.random((H, W))),
                        0.6 + 4 * rng.random((H, W))))
    params[0, 0] = (3.0, 0.2, 0.7)             # a line on the first channels: wraps where the grid does
    params[0, 1] = (2.0, D - 1.1, 0.9)         # ... and on the last
    params[1, 0] = (4.0, 5.0, 0.0)             # w == 0: the delta at z == c
    mask = np.ones((H, W), dtype=np.uint8)
    mask[2, 3] = mask[6, 8] = 0
    clean0, conv0 = line_cubes(D, H, W, lsf, params, mask, 0)
    clean1, conv1 = line_cubes(D, H, W, lsf, params, mask, 1)
    clean2, conv2 = line_cubes(D, H, W, lsf, params, mask, 2)
    np.testing.assert_array_equal(clean1, clean0)
    np.testing.assert_array_equal(conv1, conv0)
    for rounds in (3, 8):                      # several spaxel rounds per wavefront, ragged last one
        clean_r, conv_r = line_cubes(D, H, W, lsf, params, mask, 1, rounds)
        np.testing.assert_array_equal(clean_r, clean0)
        np.testing.assert_array_equal(conv_r, conv0)
    clean_r, conv_r = line_cubes(D, H, W, lsf, params, mask, 2, 5)
    np.testing.assert_array_equal(clean_r, clean2)
    np.testing.assert_array_equal(conv_r, conv2)
    scale = np.max(np.abs(clean0))
    assert np.max(np.abs(clean2 - clean0)) <= 4e-16 * scale
    assert np.max(np.abs(conv2 - conv0)) <= 4e-16 * scale
    assert (clean2[:, mask == 0] == 0).all() and (conv2[:, mask == 0] == 0).all()
    want_clean = O.simulate_clean((D, H, W), params, mask)
    want = O.lsf_lines((D, H, W), params, mask, lsf) if lsf is not None else want_clean
    ok = np.ones((H, W), dtype=bool)
    ok[1, 0] = False                           # (the reference divides 0 / 0 there)
    assert np.max(np.abs(clean2 - want_clean)[:, ok]) <= 1e-14 * scale
    assert np.max(np.abs(conv2 - want)[:, ok]) <= 1e-13 * scale
