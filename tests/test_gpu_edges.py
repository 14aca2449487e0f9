"""
GPU edge cases of the hot path (through the C ABI): NaN voxels and zero
variance (lib/run.py:153-165, 174-182), deep cubes (every MH kernel geometry),
the depth limit, degenerate widths, masks that empty a colour class.
"""
import numpy as np
import pytest

from deconv3d_amd import _lib
from oracle import deconv3d_oracle as O

pytestmark = pytest.mark.gpu


def small_problem(D, H, W, fsf, lsf, seed=0):
    rng = np.random.default_rng(seed)
    truth = np.dstack((1 + 5 * rng.random((H, W)), D * (0.3 + 0.4 * rng.random((H, W))),
                       1.0 + 2 * rng.random((H, W))))
    mask = np.ones((H, W))
    clean = O.forward_full((D, H, W), truth, mask, fsf, lsf)
    sigma = 0.05 * clean.max()
    data = clean + rng.normal(0, sigma, clean.shape)
    var = np.full(clean.shape, sigma ** 2)
    min_b = O.model_min_boundaries()
    max_b = O.model_max_boundaries(data, fsf)
    init = min_b + (max_b - min_b) * rng.random((H, W, 3))
    init[..., 2] = np.maximum(init[..., 2], 0.5)
    return data, var, mask, truth, init, min_b, max_b


@pytest.mark.parametrize("D,lsf_kind", [(d, "gauss") for d in (100, 128, 200, 256, 257, 300, 512, 513, 1000,
                                                                 1024, 1025, 2048, 3700)] +
                         [(d, "muse") for d in (513, 770, 1024, 1025, 2048, 3700)] +
                         [(d, "gauss1.0") for d in (128, 600, 1500)])
def test_deep_cubes_chain_matches_oracle(D, lsf_kind):
    """Depths that select every MH kernel: wave-specialised with 256 (D <= 256) or 512
    streaming threads (D <= 512), plain deferred beyond, 256/512/1024-thread blocks, the z-blocked forms beyond 1024
    channels (a full MUSE cube has ~3700; lib/convolution.py:137-141 takes any depth),
    and non-power-of-two depths with the partial-wrap LSF.  "muse": LSF taps within +-8
    channels -- beyond 512 channels the z-blocked form of the wave-specialised kernel
    (k_mh_ws on 256-channel blocks + k_mh_zdecide); "gauss": a Gaussian's long tail of tiny
    taps -- the plain deferred / thread-looped kernels there.  "gauss1.0" (round 4): a Gaussian
    LSF of sigma 1.0 px, FWHM 2.94 A at MUSE's 1.25 A per channel -- the upper end of the
    instrument's range.  Its taps fit the +-8 channels of the fused and z-blocked kernels once
    the cut is an error bound of the sum (1e-16) instead of 1e-20 of the largest tap."""
    H, W = 5, 6
    fsf = O.gaussian_fsf_image(1.6)
    lsf = {"gauss": lambda d: O.gaussian_lsf_vector(d, 1.1), "gauss1.0": lambda d: O.gaussian_lsf_vector(d, 1.0),
           "muse": O.muse_like_lsf}[lsf_kind](D)
    data, var, mask, truth, init, min_b, max_b = small_problem(D, H, W, fsf, lsf, seed=D)
    st = O.MHState(data, var, mask, fsf, lsf, init, min_b, max_b, seed=3)
    with _lib.Engine((D, H, W), fsf.shape) as eng:
        if lsf_kind == "gauss1.0":
            eng.set_taps(fsf, lsf, lsf_rel_threshold=1e-20)      # round 3's cut: 9.6 sigma
            assert eng.get_option("lsf_fits") == 0
        eng.set_taps(fsf, lsf)
        assert eng.get_option("lsf_fits") == (0 if lsf_kind == "gauss" else 1)
        eng.set_data(data, var, mask=mask)
        eng.set_params(truth)
        sim = eng.forward()
        ref = O.forward_full((D, H, W), truth, mask, fsf, lsf)
        assert np.max(np.abs(sim - ref)) <= 1e-12 * np.max(np.abs(ref))
        eng.set_params(init)
        eng.mh_config(min_b, max_b, 0.1, st.ra, seed=3, refresh_every=0)
        eng.mh_sweeps(2, 1)
        for s in (1, 2):
            O.mh_sweep(st, s)
        np.testing.assert_allclose(eng.get_params(), st.params, rtol=1e-9, atol=1e-9)
        err = eng.download_slot(_lib.SLOT_ERR)
        assert np.max(np.abs(err - st.err)) <= 1e-11 * np.max(np.abs(st.err))


@pytest.mark.parametrize("D", [300, 512])
def test_512_thread_wave_specialised_kernel_is_bit_identical_to_the_other_schemes(D):
    """257 .. 512 channels: k_mh_ws with 512 streaming threads -- one or two pending layers,
    the few-windows form (four positions in flight) as chosen for this small cube -- against
    the plain deferred kernel and the immediate one (same position groups: same bits)."""
    H, W = 9, 13
    fsf = O.gaussian_fsf_image(1.6)
    lsf = O.gaussian_lsf_vector(D, 1.1)
    data, var, mask, truth, init, min_b, max_b = small_problem(D, H, W, fsf, lsf, seed=D + 1)
    outs = []
    for opts in ({"mh_defer": 1}, {"mh_defer": 1, "mh_layers": 2}, {"mh_defer": 2}, {"mh_defer": 0}):
        with _lib.Engine((D, H, W), fsf.shape, options=opts) as eng:
            eng.set_taps(fsf, lsf)
            eng.set_data(data, var, mask=mask)
            eng.set_params(init)
            eng.mh_config(min_b, max_b, 0.1, 40.0, seed=3, refresh_every=0)
            if opts == {"mh_defer": 1, "mh_layers": 2}:
                assert eng.mh_layers() == 2
            acc = eng.mh_sweeps(3, 1)
            outs.append((eng.get_params(), eng.download_slot(_lib.SLOT_ERR), eng.get_dlog(), np.int64(acc)))
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("D,uniform", [(600, False), (1500, False), (1030, True)])
def test_z_blocked_sweep_kernels_against_the_plain_ones(D, uniform):
    """Cubes deeper than 512 channels: k_mh_ws on (window, 256-channel block) workgroups +
    k_mh_zdecide (option mh_zblocks, default) -- one and two pending layers, per-voxel and
    uniform variance, masked spaxels, a ragged last block -- against the plain kernels
    (mh_zblocks = 0: k_mh_defer / k_mh_deep): another grouping of the channel sums, so to
    rounding; the two layer depths of the z-blocked form bit for bit."""
    H, W = 13, 12
    fsf = O.gaussian_fsf_image(1.6)
    lsf = O.muse_like_lsf(D)             # taps within +-8 channels: what the z-blocked form takes
    data, var, mask, truth, init, min_b, max_b = small_problem(D, H, W, fsf, lsf, seed=D + 2)
    if uniform:
        var = np.full_like(var, 0.7)
    outs = []
    for opts in ({"mh_zblocks": 1}, {"mh_zblocks": 1, "mh_layers": 2}, {"mh_zblocks": 0}):
        with _lib.Engine((D, H, W), fsf.shape, options=opts) as eng:
            eng.set_taps(fsf, lsf)
            eng.set_data(data, var, mask=mask)
            eng.set_params(init)
            eng.mh_config(min_b, max_b, 0.1, 40.0, seed=3, refresh_every=0)
            if uniform:
                assert eng.variance_is_uniform()
            assert eng.get_option("mh_zblocks") == opts["mh_zblocks"]
            # (0: immediate write-back -- the thread-looped kernel of cubes beyond 1024 channels)
            want = 2 if opts.get("mh_layers") == 2 else 1
            assert eng.mh_layers() == (want if opts["mh_zblocks"] or D <= 1024 else 0)
            acc = eng.mh_sweeps(3, 1)
            outs.append((eng.get_params(), eng.download_slot(_lib.SLOT_ERR), eng.get_dlog(), np.int64(acc)))
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a, b)
    live = mask == 1
    np.testing.assert_allclose(outs[0][0][live], outs[2][0][live], rtol=1e-9, atol=1e-9)
    assert np.max(np.abs(outs[0][1] - outs[2][1])) <= 1e-11 * np.max(np.abs(outs[2][1]))
    assert outs[0][3] == outs[2][3]
    np.testing.assert_array_equal(outs[0][0][~live], init[~live])


@pytest.mark.parametrize("D,H,W", [(700, 30, 28), (2100, 14, 25)])
def test_z_blocked_chain_carries_its_residual(D, H, W):
    """lib/run.py:521-534 on the z-blocked kernels: after 20 sweeps, split over three calls with
    a periodic from-scratch residual in between, the carried residual equals the one rebuilt
    from the parameters (only ~1e-14 of creep), with and without the refresh."""
    fsf = O.gaussian_fsf_image(1.6)
    lsf = O.muse_like_lsf(D)
    data, var, mask, truth, init, min_b, max_b = small_problem(D, H, W, fsf, lsf, seed=5)
    outs = []
    for every in (0, 7):
        with _lib.Engine((D, H, W), fsf.shape, options={"mh_layers": 2}) as eng:
            eng.set_taps(fsf, lsf)
            eng.set_data(data, var, mask=mask)
            eng.set_params(init)
            eng.mh_config(min_b, max_b, 0.1, 40.0, seed=11, refresh_every=every)
            eng.mh_sweeps(6, 1)
            eng.mh_sweeps(9, 7)
            eng.mh_sweeps(5, 16)
            carried = eng.download_slot(_lib.SLOT_ERR)
            fresh = eng.residual()
            assert np.max(np.abs(carried - fresh)) <= 1e-11 * np.max(np.abs(fresh))
            outs.append(eng.get_params())
    np.testing.assert_allclose(outs[0], outs[1], rtol=1e-8, atol=1e-8)


def test_depth_limit_is_reported():
    with pytest.raises(NotImplementedError, match="8192"):
        _lib.Engine((8193, 4, 4), (3, 3))


def test_deep_cube_probe_convolution_and_run():
    """The other entry points on a 2500-channel cube: window probe, LSF (x) FSF of an
    arbitrary cube in both layouts, chi2 map -- against the oracle."""
    D, H, W = 2500, 4, 5
    fsf = O.gaussian_fsf_image(1.2)
    lsf = O.gaussian_lsf_vector(D, 1.4)
    data, var, mask, truth, init, min_b, max_b = small_problem(D, H, W, fsf, lsf, seed=7)
    with _lib.Engine((D, H, W), fsf.shape) as eng:
        eng.set_taps(fsf, lsf)
        eng.set_data(data, var, mask=mask)
        eng.set_params(init)
        err = eng.residual()
        ref = O.compute_error_in_one_step(data, init, mask, fsf, lsf)
        assert np.max(np.abs(err - ref)) <= 1e-12 * np.max(np.abs(ref))
        p_new = init[2, 3] + np.array([0., 0.7, 0.1])
        got = eng.window_stats(2, 3, p_new)
        want = O.window_stats(err, var, init[2, 3], p_new, 2, 3, fsf, lsf)
        np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-12 * want[0])
        cube = np.random.default_rng(1).normal(size=(D, H, W))
        conv = O.convolve_cube(cube, fsf, lsf)
        assert np.max(np.abs(eng.convolve(cube) - conv)) <= 1e-12 * np.max(np.abs(conv))
        cmap, total = eng.chi2_map()
        np.testing.assert_allclose(cmap, 0.5 * np.sum(err ** 2 / var, axis=0), rtol=1e-10)


def test_z_blocked_context_between_sweeps_probe_and_chi2_flush_the_pending_layers():
    """A 700-channel context (z-blocked sweep kernels, two pending layers forced): the window
    probe and the chi2 map in the middle of a chain see the residual with every pending update
    applied, and the chain goes on as if nothing had happened -- all against the oracle."""
    D, H, W = 700, 8, 9
    fsf = O.gaussian_fsf_image(1.6)
    lsf = O.muse_like_lsf(D)
    data, var, mask, truth, init, min_b, max_b = small_problem(D, H, W, fsf, lsf, seed=21)
    st = O.MHState(data, var, mask, fsf, lsf, init, min_b, max_b, seed=4)
    with _lib.Engine((D, H, W), fsf.shape, options={"mh_layers": 2}) as eng:
        eng.set_taps(fsf, lsf)
        eng.set_data(data, var, mask=mask)
        eng.set_params(init)
        eng.mh_config(min_b, max_b, 0.1, st.ra, seed=4, refresh_every=0)
        assert eng.mh_layers() == 2
        eng.mh_sweeps(2, 1)
        for s_ in (1, 2):
            O.mh_sweep(st, s_)
        y, x = np.argwhere(mask == 1)[3]
        p_new = st.params[y, x] + np.array([0., 0.9, 0.15])
        got = eng.window_stats(int(y), int(x), p_new)
        want = O.window_stats(st.err, var, st.params[y, x], p_new, int(y), int(x), fsf, lsf)
        np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-11 * abs(want[0]))
        cmap, total = eng.chi2_map()
        np.testing.assert_allclose(cmap, 0.5 * np.sum(st.err ** 2 / var, axis=0), rtol=1e-9)
        eng.mh_sweeps(1, 3)
        O.mh_sweep(st, 3)
        np.testing.assert_allclose(eng.get_params(), st.params, rtol=1e-9, atol=1e-9)
        err = eng.download_slot(_lib.SLOT_ERR)
        assert np.max(np.abs(err - st.err)) <= 1e-11 * np.max(np.abs(st.err))


def test_nan_voxels_and_zero_variance():
    """NaN voxels get zero weight (nansum of lib/run.py:423-424), a spaxel with a
    NaN anywhere in its spectrum is never iterated (lib/run.py:159-162), zero
    variance becomes 1e12 (lib/run.py:180)."""
    D, H, W = 32, 10, 11
    fsf = O.gaussian_fsf_image(2.0)
    lsf = O.gaussian_lsf_vector(D, 0.8)
    data, var, mask, truth, init, min_b, max_b = small_problem(D, H, W, fsf, lsf, seed=1)
    data[5, 4, 6] = np.nan
    data[:, 0, 0] = np.nan
    var[7, 2, 3] = 0.0
    with _lib.Engine((D, H, W), fsf.shape) as eng:
        eng.set_taps(fsf, lsf)
        eng.set_data(data, var, mask=mask)
        eng.set_params(init)
        eng.mh_config(min_b, max_b, 0.1, 50.0, seed=2, refresh_every=0)
        err = eng.residual()
        assert np.isfinite(err).all()
        model = O.forward_full((D, H, W), init, _nanmask(mask, data), fsf, lsf)
        assert abs(err[5, 4, 6] + model[5, 4, 6]) <= 1e-12 * np.max(np.abs(model))   # data := 0 there
        # oracle with the same conventions: masked NaN spaxels, data 0 / huge variance there
        omask = _nanmask(mask, data)
        odata = np.where(np.isnan(data), 0.0, data)
        ovar = np.where(var == 0.0, 1e12, var)
        ovar = np.where(np.isnan(data), np.inf, ovar)
        cmap, total = eng.chi2_map()
        ref_map = O.chi2_map(odata - O.forward_full((D, H, W), init, omask, fsf, lsf), ovar)
        np.testing.assert_allclose(cmap, ref_map, rtol=1e-10, atol=1e-12 * ref_map.sum())
        st = O.MHState(odata, ovar, omask, fsf, lsf, init, min_b, max_b, gibbs_apriori_variance=50.0,
                       seed=2)
        eng.mh_sweeps(2, 1)
        for s in (1, 2):
            O.mh_sweep(st, s)
        p = eng.get_params()
        assert np.isfinite(p).all()
        np.testing.assert_allclose(p, st.params, rtol=1e-9, atol=1e-9)
        np.testing.assert_array_equal(p[4, 6], init[4, 6])      # NaN spectrum: never iterated
        np.testing.assert_array_equal(p[0, 0], init[0, 0])


def _nanmask(mask, data):
    m = np.array(mask, dtype=float)
    m[np.isnan(np.sum(data, 0))] = 0
    return m


def test_mask_that_empties_colour_classes_and_single_live_spaxel():
    D, H, W = 16, 9, 9
    fsf = O.gaussian_fsf_image(2.0)          # 7x7 -> 49 colours, 81 spaxels
    lsf = O.gaussian_lsf_vector(D, 0.7)
    data, var, mask, truth, init, min_b, max_b = small_problem(D, H, W, fsf, lsf, seed=2)
    mask[:] = 0
    mask[4, 4] = 1
    mask[8, 0] = 1
    st = O.MHState(data, var, mask, fsf, lsf, init, min_b, max_b, seed=6)
    with _lib.Engine((D, H, W), fsf.shape) as eng:
        eng.set_taps(fsf, lsf)
        eng.set_data(data, var, mask=mask)
        eng.set_params(init)
        eng.mh_config(min_b, max_b, 0.1, st.ra, seed=6, refresh_every=0)
        assert sum(eng.colour_count(c) for c in range(49)) == 2
        eng.mh_sweeps(3, 1)
        for s in (1, 2, 3):
            O.mh_sweep(st, s)
        np.testing.assert_allclose(eng.get_params(), st.params, rtol=1e-9, atol=1e-9)
        err = eng.download_slot(_lib.SLOT_ERR)
        assert np.max(np.abs(err - st.err)) <= 1e-11 * np.max(np.abs(st.err))
    # everything masked: nothing to do, nothing breaks
    mask[:] = 0
    with _lib.Engine((D, H, W), fsf.shape) as eng:
        eng.set_taps(fsf, lsf)
        eng.set_data(data, var, mask=mask)
        eng.set_params(init)
        eng.mh_config(min_b, max_b, 0.1, 10.0, seed=6, refresh_every=0)
        assert eng.mh_sweeps(2, 1) == 0
        np.testing.assert_array_equal(eng.get_params(), init)


def test_zero_width_line_is_a_delta_not_nan():
    """w = 0 is inside the reference's bounds (lib/line_models.py:77) and gives
    0/0 there (:109); the device builds a delta line at z == c instead."""
    D, H, W = 16, 4, 4
    fsf = np.ones((1, 1))
    data = np.zeros((D, H, W))
    params = np.zeros((H, W, 3))
    params[..., 2] = 1.0
    params[1, 2] = [3.0, 5.0, 0.0]
    with _lib.Engine((D, H, W), fsf.shape) as eng:
        eng.set_taps(fsf, None)
        eng.set_data(data + 1.0, None, 1.0)
        eng.set_params(params)
        clean = eng.build_clean()
    assert np.isfinite(clean).all()
    want = np.zeros(D)
    want[5] = 3.0
    np.testing.assert_array_equal(clean[:, 1, 2], want)


def test_refresh_cadence_inside_mh_sweeps():
    """refresh_every (lib/run.py:525): the from-scratch residual replaces the
    carried one without changing the chain beyond round-off."""
    D, H, W = 16, 8, 8
    fsf = O.gaussian_fsf_image(2.0)
    lsf = O.gaussian_lsf_vector(D, 0.7)
    data, var, mask, truth, init, min_b, max_b = small_problem(D, H, W, fsf, lsf, seed=3)
    outs = []
    for every in (0, 2):
        with _lib.Engine((D, H, W), fsf.shape) as eng:
            eng.set_taps(fsf, lsf)
            eng.set_data(data, var, mask=mask)
            eng.set_params(init)
            eng.mh_config(min_b, max_b, 0.1, 30.0, seed=8, refresh_every=every)
            eng.mh_sweeps(5, 1)
            outs.append((eng.get_params(), eng.download_slot(_lib.SLOT_ERR)))
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=1e-8, atol=1e-8)
    assert np.max(np.abs(outs[0][1] - outs[1][1])) <= 1e-10 * np.max(np.abs(outs[0][1]))


def test_pending_layer_policy():
    """d3d_mh_layers: small cubes (colour launches that do not fill the chip) write
    the residual back every colour, option mh_layers forces a depth, depths beyond
    160 / 512 channels cap it at 2 / 1, tiled contexts write at once."""
    fsf = O.gaussian_fsf_image(1.6)

    def layers(D, forced=None):
        H, W = 6, 7
        data, var, mask, truth, init, min_b, max_b = small_problem(D, H, W, fsf, None, seed=1)
        with _lib.Engine((D, H, W), fsf.shape) as eng:
            eng.set_taps(fsf, None)
            eng.set_data(data, var, mask=mask)
            if forced is not None:     # after the data: the work lists are rebuilt
                eng.set_option("mh_layers", forced)
            return eng.mh_layers()

    assert layers(64) == 1
    assert layers(64, 2) == 2
    assert layers(64, 3) == 3
    assert layers(200, 3) == 2
    assert layers(300, 3) == 2          # 512 streaming threads (257 .. 512 channels)
    assert layers(600, 3) == 2          # beyond: the z-blocked form of the same kernel

