"""
k_conv_rows (csrc/d3d_conv.h): the one-pass LSF (x) FSF convolution for 128-, 64- and
32-channel cubes (and, FSF only, in z-blocks of 128 channels for every depth above 128)
and mirror-symmetric FSFs -- loader wavefront + LDS row ring + one-column
register rings + LSF epilogue -- against the oracle's restatement of
lib/convolution.py:89-120 and lib/run.py:1027-1029, and against the two-pass
kernels it replaces (option conv_rows = 0).  Shapes chosen to hit every border case of
the kernel's geometry: widths that are not a multiple of the 15 columns a workgroup
owns, strips of unequal height, fewer rows than the FSF, single columns.
"""
import numpy as np
import pytest

from deconv3d_amd import _lib
from oracle import deconv3d_oracle as O

pytestmark = pytest.mark.gpu


def engine(shape, fsf, lsf, conv_rows=True, hy=None):
    eng = _lib.Engine(shape, fsf.shape, options={"conv_rows": 1 if conv_rows else 0,
                                                 "conv_hy": hy or 0})
    eng.set_taps(fsf, lsf)
    return eng


def lsf_asym(D, rng):
    """Dense asymmetric taps within +-8 channels of the centre (lib/spread_functions.py:251)."""
    v = np.zeros(D)
    zc = (D - 1) // 2 - (D % 2 - 1)
    v[zc - 8:zc + 9] = rng.random(17)
    return v / v.sum()


CASES = [  # (D, H, W, fsf size, lsf kind, strip height override)
    (128, 23, 19, 11, "muse", None),
    (128, 31, 17, 11, "asym", 7),
    (128, 9, 47, 9, "muse", 4),
    (128, 40, 15, 7, "none", None),
    (128, 5, 16, 11, "asym", None),       # fewer rows than the FSF
    (128, 33, 1, 5, "muse", 6),           # a single column
    (128, 26, 31, 13, "asym", 9),
    (128, 12, 30, 3, "muse", None),
    (127, 14, 22, 11, "muse", None),      # padded depth: LSF not power-of-two -> FSF pass only
    # 64 / 32 channels: two / four adjacent columns per wavefront (BASELINE configs 2 and 1)
    (64, 23, 31, 11, "muse", None),
    (64, 40, 30, 11, "asym", 7),          # exactly one workgroup's 30 columns
    (64, 9, 61, 9, "muse", 4),            # 2 workgroups + one column
    (64, 26, 1, 11, "muse", 6),           # a single column: half a wavefront idle
    (64, 12, 29, 9, "none", None),
    (32, 16, 16, 9, "muse", None),        # config 1's footprint
    (32, 21, 63, 11, "asym", 5),          # 60 columns per workgroup + 3
    (32, 7, 2, 9, "muse", None),
    (63, 14, 22, 11, "muse", None),       # padded depth 64: FSF pass only
    # above 128 channels: the FSF pass in z-blocks of 128 channels (LSF in a pass of its own)
    (256, 23, 19, 11, "muse", None),      # two full blocks
    (200, 17, 31, 9, "asym", 5),          # ragged last block (72 channels), 2 workgroups + 1 column
    (130, 12, 16, 13, "none", None),      # last block: ONE z-pair
    (384, 9, 33, 11, "muse", 4),          # three blocks
    (301, 6, 7, 9, "muse", None),         # odd depth: padded to 302
    (1500, 5, 6, 11, "muse", None),       # a deep cube (thread-looped line / LSF kernels): 12 blocks
    # the LSF pass in 128-channel blocks (k_spectral_blocks): depths within 8 channels of the
    # power-of-two padded length wrap partially (lib/convolution.py:137-160)
    (255, 6, 5, 9, "asym", None),
    (249, 4, 9, 11, "asym", None),
    (1020, 3, 4, 9, "asym", None),
    (100, 7, 6, 9, "asym", None),         # one ragged block
    # round 4: 15 x 15 -- the footprint of the reference's own science fixture
    # (tests/read_mat.py:28-36) -- in the radial form, with the LSF epilogue and in z-blocks
    (128, 23, 19, 15, "muse", None),
    (128, 9, 33, 15, "asym", 4),          # fewer rows than the FSF
    (128, 31, 16, 15, "none", 7),
    (256, 17, 20, 15, "muse", None),
    # ... and depths below 128 that are no power of two: ONE ragged z-block (FSF pass only)
    (21, 30, 24, 15, "muse", None),       # the fixture's shape
    (30, 17, 31, 11, "asym", 5),
    (48, 9, 16, 9, "none", None),
    (2, 5, 7, 9, "none", None),
]


@pytest.mark.parametrize("D,H,W,fs,lsf_kind,hy", CASES)
def test_one_pass_convolution_matches_the_oracle_and_the_two_pass_kernels(D, H, W, fs, lsf_kind, hy):
    rng = np.random.default_rng(D * 1000 + H * 31 + W)
    fsf = O.moffat_cropped(fs, 3.0, 2.5)
    lsf = {"muse": O.muse_like_lsf, "asym": lambda d: lsf_asym(d, rng), "none": lambda d: None}[lsf_kind](D)
    cube = rng.normal(size=(D, H, W))
    want = O.convolve_cube(cube, fsf, lsf) if lsf is not None else O.spatial_convolve(cube, fsf)
    outs = []
    for conv_rows in (True, False):
        with engine((D, H, W), fsf, lsf, conv_rows, hy) as eng:
            eng.upload_slot(_lib.SLOT_TMP0, cube)
            eng.convolve_slots(_lib.SLOT_TMP0, _lib.SLOT_SIM)
            outs.append(eng.download_slot(_lib.SLOT_SIM))
    scale = np.max(np.abs(want))
    assert np.max(np.abs(outs[0] - want)) <= 1e-12 * scale
    assert np.max(np.abs(outs[1] - want)) <= 1e-12 * scale
    assert np.max(np.abs(outs[0] - outs[1])) <= 1e-13 * scale


@pytest.mark.parametrize("D,H,W,fs,hy", [(128, 21, 34, 11, 8), (128, 17, 16, 9, None),
                                         (64, 21, 34, 11, 8), (32, 17, 16, 9, None),
                                         (256, 21, 34, 11, 8), (200, 17, 16, 9, None)])
def test_forward_model_and_residual_through_the_one_pass_kernel(D, H, W, fs, hy):
    """params -> lines -> FSF, plain and with the data - sim epilogue
    (lib/run.py:999-1031), with masked spaxels."""
    rng = np.random.default_rng(7)
    fsf = O.moffat_cropped(fs, 3.0, 2.5)
    lsf = O.muse_like_lsf(D)
    params = np.dstack((1 + 9 * rng.random((H, W)), D * (0.2 + 0.6 * rng.random((H, W))),
                        0.8 + 3 * rng.random((H, W))))
    mask = np.ones((H, W))
    mask[3, 5] = mask[H - 1, W - 1] = 0
    data = rng.normal(size=(D, H, W))
    var = 0.5 + rng.random((D, H, W))
    want = O.forward_full((D, H, W), params, mask, fsf, lsf)
    with engine((D, H, W), fsf, lsf, True, hy) as eng:
        eng.set_data(data, var, mask=mask)
        eng.set_params(params)
        sim = eng.forward()
        err = eng.residual()
    assert np.max(np.abs(sim - want)) <= 1e-12 * np.max(np.abs(want))
    assert np.max(np.abs(err - (data - want))) <= 1e-12 * np.max(np.abs(data - want))


@pytest.mark.parametrize("D", [128, 64, 32, 256, 200])
def test_one_pass_kernel_is_position_independent(D):
    """The march always runs top-down: a cube cut out of a larger one (with the FSF
    half width of context) gives the very same bits on the common interior -- which is
    what lets a tile rebuild its residual and agree with the full cube."""
    rng = np.random.default_rng(11)
    H, W, fs = 44, 37, 11
    fsf = O.moffat_cropped(fs, 3.0, 2.5)
    lsf = O.muse_like_lsf(D)
    cube = rng.normal(size=(D, H, W))
    with engine((D, H, W), fsf, lsf, True, 13) as eng:
        eng.upload_slot(_lib.SLOT_TMP0, cube)
        eng.convolve_slots(_lib.SLOT_TMP0, _lib.SLOT_SIM)
        full = eng.download_slot(_lib.SLOT_SIM)
    y0, y1, x0, x1 = 9, 40, 6, 33
    with engine((D, y1 - y0, x1 - x0), fsf, lsf, True, 9) as eng:
        eng.upload_slot(_lib.SLOT_TMP0, np.ascontiguousarray(cube[:, y0:y1, x0:x1]))
        eng.convolve_slots(_lib.SLOT_TMP0, _lib.SLOT_SIM)
        part = eng.download_slot(_lib.SLOT_SIM)
    h = fs // 2
    np.testing.assert_array_equal(part[:, h:-h, h:-h], full[:, y0 + h:y1 - h, x0 + h:x1 - h])


@pytest.mark.parametrize("kind", ["radial", "elliptical", "outer product"])
@pytest.mark.parametrize("D", [256, 168])
def test_z_blocked_fsf_pass_for_every_symmetric_fsf_class(kind, D):
    """Depths above 128 (option conv_zb): each of the kernel's three tap forms -- radial
    (transposition-symmetric quadrant), mirror-symmetric only (an elliptical Gaussian with
    pa = 0, lib/spread_functions.py:113-131), outer product (a circular Gaussian) -- against
    scipy's convolve2d per channel (the oracle's spatial pass) and the march kernels
    (conv_zb = 0)."""
    rng = np.random.default_rng(3)
    H, W = 19, 33
    fsf = {"radial": O.moffat_cropped(11, 3.0, 2.5),
           "elliptical": O.gaussian_fsf_image(4.2, pa=0., ba=0.6),
           "outer product": O.gaussian_fsf_image(4.2)}[kind]
    assert fsf.shape[0] in (9, 11, 13)
    cube = rng.normal(size=(D, H, W))
    want = O.spatial_convolve(cube, fsf)
    outs = []
    for zb in (1, 0):
        with _lib.Engine((D, H, W), fsf.shape, options={"conv_zb": zb, "conv_hy": 7}) as eng:
            eng.set_taps(fsf, None)
            eng.upload_slot(_lib.SLOT_TMP0, cube)
            eng.convolve_slots(_lib.SLOT_TMP0, _lib.SLOT_SIM)
            outs.append(eng.download_slot(_lib.SLOT_SIM))
    scale = np.max(np.abs(want))
    for got in outs:
        assert np.max(np.abs(got - want)) <= 1e-12 * scale
    assert not np.array_equal(outs[0], outs[1])        # two kernels did run


def test_reference_fixture_forward_model_runs_the_one_pass_kernel():
    """tests/input/data14forAntoine.mat (committed as golden/ref_mat_fixture.npz): 21 channels,
    24 x 30 spaxels, the 15 x 15 FSF of the file itself (radial to the last bit).  Round 3's
    k_conv_rows took neither the footprint nor the depth; now the forward model of the
    theoretical parameters (c - 1: tests/read_mat.py:49-68) equals the oracle's and is the same
    with the one-pass kernel switched off."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_mat_fixture.npz"))
    data, fsf, params = g["data"], np.ascontiguousarray(g["fsf"]), g["params"].copy()
    D, H, W = data.shape
    assert fsf.shape == (15, 15) and np.array_equal(fsf, fsf.T) and np.array_equal(fsf, fsf[::-1])
    want = O.forward_full((D, H, W), params, np.ones((H, W)), fsf, None)
    outs = []
    for conv_rows in (True, False):
        with engine((D, H, W), fsf, None, conv_rows) as eng:
            eng.set_params(params)
            outs.append(eng.forward())
    scale = np.max(np.abs(want))
    assert np.max(np.abs(outs[0] - want)) <= 1e-12 * scale
    assert np.max(np.abs(outs[0] - outs[1])) <= 1e-13 * scale
