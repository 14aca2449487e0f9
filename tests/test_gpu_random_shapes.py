"""
Randomised shapes through the C ABI against the oracle: depth, footprint, FSF
size and symmetry class, LSF kind, mask density, NaN voxels and the variance
kind are drawn per seed, so that every kernel selection rule of the library
(tile / march / separable / generic spatial pass, dense / tap-list LSF pass,
wave-specialised / plain deferred MH kernel, uniform-variance variant, one or
four window positions in flight) meets shapes nobody picked by hand.
Tolerances as in tests/test_gpu_parity.py.
"""
import numpy as np
import pytest

from deconv3d_amd import _lib
from oracle import deconv3d_oracle as O

pytestmark = pytest.mark.gpu


def draw_case(seed, depth=None, extent=34):
    rng = np.random.default_rng(1000 + seed)
    D = int(rng.choice([1, 2, 3, 7, 16, 17, 31, 32, 33, 64, 65, 100, 128, 130]))
    if depth is not None:
        D = depth
    H, W = int(rng.integers(1, extent)), int(rng.integers(1, extent))
    kind = rng.choice(["gauss", "ellipse", "rotated", "moffat", "random", "rect", "delta"])
    if kind == "gauss":
        fsf = O.gaussian_fsf_image(float(rng.uniform(0.8, 4.2)))
    elif kind == "ellipse":
        fsf = O.gaussian_fsf_image(float(rng.uniform(1.5, 4.0)), pa=0., ba=float(rng.uniform(0.4, 0.9)))
    elif kind == "rotated":
        fsf = O.gaussian_fsf_image(float(rng.uniform(1.5, 4.0)), pa=float(rng.uniform(10, 80)), ba=0.6)
    elif kind == "moffat":
        fsf = O.moffat_cropped(int(rng.choice([3, 5, 7, 9, 11, 13])), float(rng.uniform(1.5, 3.5)), 2.5)
    elif kind == "random":
        n = int(rng.choice([3, 5, 7]))
        fsf = rng.random((n, n))
        fsf /= fsf.sum()
    elif kind == "rect":
        fsf = rng.random((int(rng.choice([1, 3, 5])), int(rng.choice([3, 5, 7, 9]))))
        fsf /= fsf.sum()
    else:
        fsf = np.ones((1, 1))
    lkind = rng.choice(["none", "gauss", "wide", "dense"] + (["muse", "muse"] if D > 130 else []))
    if lkind == "none":
        lsf = None
    elif lkind == "muse":
        lsf = O.muse_like_lsf(D)                                          # taps within +-8 channels
    elif lkind == "gauss":
        lsf = O.gaussian_lsf_vector(D, float(rng.uniform(0.3, 1.2)))
    elif lkind == "wide":
        lsf = O.gaussian_lsf_vector(D, float(rng.uniform(2.0, 4.0)))     # beyond +-8 channels
    else:
        lsf = rng.random(D)
        lsf /= lsf.sum()
    truth = np.dstack((1.0 + 9.0 * rng.random((H, W)), D * (0.2 + 0.6 * rng.random((H, W))),
                       0.6 + 2.0 * rng.random((H, W))))
    mask = (rng.random((H, W)) < rng.choice([1.0, 0.9, 0.5])).astype(float)
    if mask.sum() == 0:
        mask[rng.integers(0, H), rng.integers(0, W)] = 1
    clean = O.forward_full((D, H, W), truth, mask, fsf, lsf)
    sigma = 0.05 * np.max(clean) + 1e-3
    data = clean + rng.normal(0., sigma, size=(D, H, W))
    vkind = rng.choice(["cube", "uniform", "scalar"])
    var = (sigma * (0.5 + rng.random((D, H, W)))) ** 2 if vkind == "cube" else np.full((D, H, W), sigma ** 2)
    if rng.random() < 0.3 and D * H * W > 8:
        for _ in range(3):
            data[rng.integers(0, D), rng.integers(0, H), rng.integers(0, W)] = np.nan
    min_b = O.model_min_boundaries()
    max_b = O.model_max_boundaries(np.nan_to_num(data), fsf)
    init = min_b + (max_b - min_b) * rng.random((H, W, 3))
    init[..., 2] = np.maximum(init[..., 2], 0.3)
    return dict(D=D, H=H, W=W, fsf=fsf, lsf=lsf, truth=truth, mask=mask, data=data, var=var,
                vkind=vkind, min_b=min_b, max_b=max_b, init=init,
                what="%s fsf %s, %s lsf, %s variance" % (kind, fsf.shape, lkind, vkind))


def test_random_shape_at_depth_128_matches_oracle(seed128):
    """The 128-channel kernels (k_conv_rows: one wavefront per spectrum, 15 columns per
    workgroup) on drawn footprints up to 49x49, i.e. up to four column groups."""
    check_case(draw_case(200 + seed128, depth=128 if seed128 % 4 else 127, extent=50), seed128)


@pytest.fixture(params=range(10))
def seed128(request):
    return request.param


@pytest.mark.parametrize("seed,depth", [(s, d) for s, d in enumerate(
    [200, 264, 300, 512, 520, 700, 770, 1030, 1100, 2100, 257, 640])])
def test_random_shape_at_greater_depths_matches_oracle(seed, depth):
    """Depths beyond 128 channels: the z-blocked FSF pass, the LSF pass in 128-channel blocks, the
    512-thread sweep kernel (257 .. 512 channels) and, for LSF taps within +-8 channels, the
    z-blocked sweep kernels beyond -- on drawn footprints, FSF classes, LSF kinds, masks and
    variances."""
    check_case(draw_case(900 + seed, depth=depth, extent=22), seed)


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("D3D_TEST_RANDOM_SHAPES", "32"))))
def test_random_shape_matches_oracle(seed):
    check_case(draw_case(seed), seed)


def check_case(c, seed):
    D, H, W = shape = (c["D"], c["H"], c["W"])
    # the oracle (like lib/run.py:153-162) drops spaxels with a NaN in their spectrum
    nan_spax = np.isnan(c["data"]).any(axis=0)
    mask = c["mask"] * (~nan_spax)
    data = c["data"]
    var = c["var"]
    with _lib.Engine(shape, c["fsf"].shape) as eng:
        eng.set_taps(c["fsf"], c["lsf"])
        if c["vkind"] == "scalar":
            eng.set_data(data, None, var_scalar=float(var.flat[0]), mask=c["mask"])
        else:
            eng.set_data(data, var, mask=c["mask"])
        assert eng.variance_is_uniform() == (c["vkind"] != "cube" and not nan_spax.any())
        eng.set_params(c["truth"])
        sim = eng.forward()
        ref = O.forward_full(shape, c["truth"], mask, c["fsf"], c["lsf"])
        assert np.max(np.abs(sim - ref)) <= 1e-12 * max(np.max(np.abs(ref)), 1e-300), c["what"]
        cube = np.random.default_rng(seed).normal(size=shape)
        out = eng.convolve(cube)
        refc = O.convolve_cube(cube, c["fsf"], c["lsf"])
        assert np.max(np.abs(out - refc)) <= 1e-12 * np.max(np.abs(refc)), c["what"]
        eng.upload_slot(_lib.SLOT_TMP0, cube)             # and between two device slots
        eng.convolve_slots(_lib.SLOT_TMP0, _lib.SLOT_SIM)
        out = eng.download_slot(_lib.SLOT_SIM)
        assert np.max(np.abs(out - refc)) <= 1e-12 * np.max(np.abs(refc)), c["what"]
        if mask.sum() == 0:
            return
        # NaN voxels: data 0, 1/var 0 there, as the device prepares them
        d0 = np.where(np.isnan(data), 0.0, data)
        v0 = np.where(np.isnan(data), np.inf, var)
        st = O.MHState(d0, v0, mask, c["fsf"], c["lsf"], c["init"], c["min_b"], c["max_b"],
                       jump_amplitude=0.1, seed=50 + seed)
        eng.set_params(c["init"])
        eng.mh_config(c["min_b"], c["max_b"], 0.1, st.ra, seed=50 + seed, refresh_every=0)
        accepted = eng.mh_sweeps(2, 1)
        for s in (1, 2):
            O.mh_sweep(st, s)
        live = mask == 1
        np.testing.assert_allclose(eng.get_params()[live], st.params[live], rtol=1e-9, atol=1e-9,
                                   err_msg=c["what"])
        assert accepted == st.accepted, c["what"]
        err = eng.download_slot(_lib.SLOT_ERR)
        assert np.max(np.abs(err - st.err)) <= 1e-11 * max(np.max(np.abs(st.err)), 1e-300), c["what"]
