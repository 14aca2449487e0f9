"""
GPU parity tests proper: the HIP path, called through the C ABI
(deconv3d_amd._lib.Engine -> libdeconv3d_hip.so), against the CPU oracle on
the same seeded inputs.  fp64 tolerances:
  convolved cube / residual : max|d| <= 1e-12 * max|cube|
  1/2 chi2, delta, moments  : rel <= 1e-10 with abs floor 1e-12 * sum
"""
import numpy as np
import pytest

from oracle import deconv3d_oracle as O
from tests.cases import ALL_CASES, make_case

pytestmark = pytest.mark.gpu

CUBE_RTOL = 1e-12


def engine_for(case, options=None):
    from deconv3d_amd import _lib
    D, H, W = case["D"], case["H"], case["W"]
    eng = _lib.Engine((D, H, W), case["fsf"].shape, options=options)
    eng.set_taps(case["fsf"], case["lsf"])
    eng.set_data(case["data"], case["var"], mask=case["mask"])
    return eng


def assert_cube_close(a, b, what):
    scale = max(np.max(np.abs(b)), 1e-300)
    err = np.max(np.abs(a - b))
    assert err <= CUBE_RTOL * scale, "%s: max|d|=%g vs scale %g" % (what, err, scale)


@pytest.mark.parametrize("name", ALL_CASES)
def test_forward_residual_chi2(name):
    case = make_case(name)
    shape = (case["D"], case["H"], case["W"])
    with engine_for(case) as eng:
        eng.set_params(case["truth"])
        sim = eng.forward()
        ref = O.forward_full(shape, case["truth"], case["mask"], case["fsf"], case["lsf"])
        assert_cube_close(sim, ref, "forward")
        # == Run.simulate_convolved (sum of contributions), lib/run.py:623-652
        if shape[1] * shape[2] <= 300:
            ref2 = O.simulate_convolved(shape, case["truth"], case["mask"], case["fsf"], case["lsf"])
            assert_cube_close(sim, ref2, "simulate_convolved")
        clean = eng.build_clean()
        assert_cube_close(clean, O.simulate_clean(shape, case["truth"], case["mask"]), "clean")
        eng.set_params(case["init"])
        err = eng.residual()
        ref_err = O.compute_error_in_one_step(case["data"], case["init"], case["mask"],
                                              case["fsf"], case["lsf"])
        assert_cube_close(err, ref_err, "residual")
        cmap, total = eng.chi2_map()
        ref_map = O.chi2_map(ref_err, case["var"])
        np.testing.assert_allclose(cmap, ref_map, rtol=1e-10, atol=1e-12 * ref_map.sum())
        np.testing.assert_allclose(total, ref_map.sum(), rtol=1e-10)


@pytest.mark.parametrize("name", ALL_CASES)
def test_convolve_arbitrary_cube(name):
    case = make_case(name)
    rng = case["rng"]
    cube = rng.normal(size=(case["D"], case["H"], case["W"]))
    with engine_for(case) as eng:
        out = eng.convolve(cube)
    assert_cube_close(out, O.convolve_cube(cube, case["fsf"], case["lsf"]), "convolve")


@pytest.mark.parametrize("kind", ["gaussian", "elliptical", "outer product of two ramps"])
@pytest.mark.parametrize("shape", [(128, 40, 37), (30, 21, 50)])
def test_outer_product_fsf_uses_the_separable_pass(kind, shape):
    """An FSF that is u v^T to rounding (every Gaussian with pa = 0) runs the
    spatial pass as 2*FS taps (k_spatial_sep).  Against the oracle's 2-D sum and
    against the device's own 2-D kernel (option spatial_sep = 0): rounding only."""
    from deconv3d_amd import _lib
    from deconv3d_amd.spread_functions import gaussian_image
    rng = np.random.default_rng(3)
    if kind == "gaussian":
        fsf = gaussian_image(4.0)
    elif kind == "elliptical":
        fsf = gaussian_image(3.2, pa=0., ba=0.6)
    else:
        fsf = np.outer(np.linspace(0.2, 1.4, 7), np.linspace(2.0, 0.5, 7))
        fsf /= fsf.sum()
    assert fsf.shape[0] == fsf.shape[1]
    D, H, W = shape
    lsf = O.gaussian_lsf_vector(D, 0.8)
    cube = rng.normal(size=shape)
    outs = []
    # separable with the LSF in the same pass (power-of-two depths), separable
    # after the streaming LSF pass, 2-D kernel; each also through the forward
    # model, whose lines are already LSF-convolved (never the one-pass kernel)
    for sep, fuse in ((1, 1), (1, 0), (0, 0)):
        with _lib.Engine(shape, fsf.shape, options={"spatial_sep": sep, "sep_fuse": fuse}) as eng:
            eng.set_taps(fsf, lsf)
            eng.upload_slot(_lib.SLOT_TMP0, cube)
            eng.convolve_slots(_lib.SLOT_TMP0, _lib.SLOT_SIM)
            outs.append(eng.download_slot(_lib.SLOT_SIM))
            eng.set_data(cube, None, var_scalar=1.0)
            eng.set_params(np.tile([1.0, D / 2.0, 1.5], (H, W, 1)))
            outs.append(eng.forward())
    ref = O.convolve_cube(cube, fsf, lsf)
    for k, what in ((0, "one pass"), (2, "separable"), (4, "2-D")):
        assert_cube_close(outs[k], ref, what + " vs oracle")
        assert np.max(np.abs(outs[k] - outs[4])) <= 1e-13 * np.max(np.abs(ref))
        assert np.max(np.abs(outs[k + 1] - outs[5])) <= 1e-13 * np.max(np.abs(outs[5]))
    assert not np.array_equal(outs[2], outs[4])      # two different summation orders did run
    if D & (D - 1) == 0:
        assert not np.array_equal(outs[0], outs[2])  # and the one-pass form too


def test_spectral_pass_by_wavefront_shuffles_is_bit_identical():
    """Option spectral_shfl = 1 (EXPERIMENTS build): the dense LSF pass exchanges neighbouring channels
    with wavefront shuffles instead of the wave-private LDS window (depth 128: one
    spectrum per wavefront).  Same taps, same order: same bits."""
    from deconv3d_amd import _lib
    if not _lib.has_experiments():
        pytest.skip("k_spectral_shfl is only in a `make EXPERIMENTS=1` build")
    shape = (128, 9, 11)
    rng = np.random.default_rng(8)
    cube = rng.normal(size=shape)
    lsf = O.muse_like_lsf(128)
    fsf = np.ones((1, 1))
    outs = []
    for knob in (0, 1):
        with _lib.Engine(shape, fsf.shape, options={"spectral_shfl": knob}) as eng:
            eng.set_taps(fsf, lsf)
            eng.upload_slot(_lib.SLOT_TMP0, cube)
            eng.convolve_slots(_lib.SLOT_TMP0, _lib.SLOT_SIM)
            outs.append(eng.download_slot(_lib.SLOT_SIM))
    np.testing.assert_array_equal(outs[0], outs[1])
    assert_cube_close(outs[1], O.convolve_cube(cube, fsf, lsf), "shuffle LSF pass vs oracle")


def test_reference_saved_cube_pair_on_device():
    """The reference's own saved pair (tests/golden/ref_galpak_pair.npz, see
    tests/test_oracle.py): the HIP convolution of its clean cube under the MUSE
    defaults reproduces its convolved cube to the same 0.4 % the oracle does, and
    equals the oracle to fp64 round-off.  Taps come from the product's MUSE
    instrument, not from the oracle."""
    import os
    from deconv3d_amd import MUSE, _lib
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_galpak_pair.npz"))
    clean, conv = g["clean"], g["convolved"]
    inst = MUSE()
    cube = inst.build_cube(clean)
    fsf, lsf = inst.fsf.as_image(cube), inst.lsf.as_vector(cube)
    with _lib.Engine(clean.shape, fsf.shape) as eng:
        eng.set_taps(fsf, lsf)
        out = eng.convolve(clean)
    assert np.abs(out - conv).max() < 5e-3 * conv.max()
    assert_cube_close(out, O.convolve_cube(clean, fsf, lsf), "convolve(saved pair)")


@pytest.mark.parametrize("name", ALL_CASES)
def test_window_stats_probe(name):
    case = make_case(name)
    rng = case["rng"]
    H, W = case["H"], case["W"]
    with engine_for(case) as eng:
        eng.set_params(case["init"])
        err = eng.residual()
        spaxels = [(0, 0), (H - 1, W - 1), (0, W - 1), (H // 2, W // 2)]
        spaxels += [(int(rng.integers(0, H)), int(rng.integers(0, W))) for _ in range(12)]
        for (y, x) in spaxels:
            p_old = case["init"][y, x]
            p_new = p_old + np.array([0., 1., 0.3]) * np.tan(np.pi * (rng.random(3) - 0.5)) * 0.5
            p_new[2] = abs(p_new[2]) + 0.2
            got = eng.window_stats(y, x, p_new)
            ref = O.window_stats(err, case["var"], p_old, p_new, y, x, case["fsf"], case["lsf"])
            floor = 1e-12 * max(ref[0], ref[1])
            np.testing.assert_allclose(got[:3], ref[:3], rtol=1e-10, atol=floor,
                                       err_msg="chi2 at %s" % ((y, x),))
            np.testing.assert_allclose(got[3:], ref[3:], rtol=1e-10,
                                       atol=1e-12 * max(abs(ref[3]), abs(ref[4])),
                                       err_msg="gibbs moments at %s" % ((y, x),))


@pytest.mark.parametrize("name", ["c1", "odd_depth", "asym", "nolsf", "rect_fsf", "tiny"])
def test_mh_chain_matches_oracle_update_by_update(name):
    """Same Philox stream, same colour order: the device chain must track the
    oracle chain (lib/run.py:367-519 restated) to fp64 round-off."""
    case = make_case(name)
    n_sweeps = 3
    st = O.MHState(case["data"], case["var"], case["mask"], case["fsf"], case["lsf"],
                   case["init"], case["min_b"], case["max_b"], jump_amplitude=0.1, seed=777)
    H, W = case["H"], case["W"]
    with engine_for(case) as eng:
        eng.set_params(case["init"])
        eng.mh_config(case["min_b"], case["max_b"], 0.1, st.ra, seed=777, refresh_every=0)
        chain = np.full((n_sweeps + 1, H, W, 3), np.nan)
        dlog = np.full((n_sweeps + 1, H, W), np.nan)
        accepted = eng.mh_sweeps(n_sweeps, 1, 1, chain, dlog)
        err_dev = eng.download_slot(2)
        for s in range(1, n_sweeps + 1):
            O.mh_sweep(st, s)
            live = case["mask"] == 1
            np.testing.assert_allclose(chain[s][live], st.params[live], rtol=1e-9, atol=1e-9,
                                       err_msg="params after sweep %d" % s)
            scale = np.max(np.abs(st.dlog[live])) + 1.0
            np.testing.assert_allclose(dlog[s][live], st.dlog[live], rtol=1e-8,
                                       atol=1e-10 * scale, err_msg="dlog sweep %d" % s)
        assert accepted == st.accepted
        assert_cube_close(err_dev, st.err, "carried residual")
        # masked spaxels never move (lib/run.py:553-566)
        dead = case["mask"] == 0
        np.testing.assert_array_equal(chain[n_sweeps][dead], case["init"][dead])


@pytest.mark.parametrize("name", ["c1", "odd_depth", "big_fsf", "tiny"])
def test_write_back_schemes_are_bit_identical(name):
    """Deferred write-back (wave-specialised k_mh_ws with three, one or two pending
    layers; k_mh_pair: two colour classes per launch with per-window G hand-off;
    k_mh_flow: one launch per sweep with per-window dependencies; plain
    k_mh_defer: the next colour applies the pending update), immediate re-read and
    immediate register-resident kernels are the same arithmetic: bit-identical
    chains and residuals."""
    case = make_case(name)
    outs = []
    from deconv3d_amd import _lib
    variants = [{"mh_defer": 1}, {"mh_defer": 1, "mh_props": 0}, {"mh_defer": 1, "mh_layers": 1},
                {"mh_defer": 1, "mh_layers": 2},
                {"mh_defer": 1, "mh_layers": 3}, {"mh_defer": 2},
                {"mh_defer": 0}]
    if _lib.has_experiments():   # k_mh_pair / k_mh_flow / register-resident k_mh: `make EXPERIMENTS=1`
        variants += [{"mh_defer": 1, "mh_layers": 2, "mh_pair": 1}, {"mh_defer": 1, "mh_flow": 1},
                     {"mh_defer": 0, "mh_maxit": 8}]   # same workgroup size: same summation order
    for opts in variants:
        with engine_for(case, options=opts) as eng:
            eng.set_params(case["init"])
            eng.mh_config(case["min_b"], case["max_b"], 0.1, 50.0, seed=5, refresh_every=0)
            eng.mh_sweeps(2, 1)
            mid = eng.download_slot(2)          # flushes the pending colour
            eng.mh_sweeps(1, 3)                 # and the chain continues consistently
            outs.append((eng.get_params(), mid, eng.download_slot(2), eng.chi2_map()[0]))
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("name", ["c1", "odd_depth", "big_fsf"])
@pytest.mark.parametrize("how", ["scalar", "constant cube"])
def test_uniform_variance_variant_is_bit_identical(name, how):
    """One constant variance (the reference's default, lib/run.py:171-178): the
    MH kernel takes 1/var from a register instead of SLOT_IVAR.  Same
    arithmetic as the general kernel (option uniform_ivar = 0) -> bit-identical
    chains, residuals and delta maps; and it matches the oracle."""
    case = make_case(name)
    shape = (case["D"], case["H"], case["W"])
    v0 = float(np.median(case["var"]))
    var_cube = np.full(shape, v0)
    outs = []
    for knob in (1, 0):
        from deconv3d_amd import _lib
        with _lib.Engine(shape, case["fsf"].shape, options={"uniform_ivar": knob}) as eng:
            eng.set_taps(case["fsf"], case["lsf"])
            if how == "scalar":
                eng.set_data(case["data"], None, var_scalar=v0, mask=case["mask"])
            else:
                eng.set_data(case["data"], var_cube, mask=case["mask"])
            assert eng.variance_is_uniform() == (knob == 1)
            eng.set_params(case["init"])
            eng.mh_config(case["min_b"], case["max_b"], 0.1, 50.0, seed=31, refresh_every=0)
            eng.mh_sweeps(2, 1)
            outs.append((eng.get_params(), eng.download_slot(2), eng.get_dlog()))
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)
    st = O.MHState(case["data"], var_cube, case["mask"], case["fsf"], case["lsf"], case["init"],
                   case["min_b"], case["max_b"], jump_amplitude=0.1, gibbs_apriori_variance=50.0,
                   seed=31)
    for s in (1, 2):
        O.mh_sweep(st, s)
    live = case["mask"] == 1
    np.testing.assert_allclose(outs[0][0][live], st.params[live], rtol=1e-9, atol=1e-9)
    assert_cube_close(outs[0][1], st.err, "carried residual (uniform variance)")


def test_uniform_variance_detection():
    """NaN voxels (1/var = 0 there), a non-constant cube, or a later upload into
    SLOT_IVAR all select the general kernel."""
    from deconv3d_amd import _lib
    case = make_case("tiny")
    shape = (case["D"], case["H"], case["W"])
    with _lib.Engine(shape, case["fsf"].shape) as eng:
        eng.set_taps(case["fsf"], case["lsf"])
        eng.set_data(case["data"], None, var_scalar=2.0)
        assert eng.variance_is_uniform()
        eng.set_data(case["data"], case["var"])
        assert not eng.variance_is_uniform()
        holed = case["data"].copy()
        holed[1, 0, 2] = np.nan
        eng.set_data(holed, None, var_scalar=2.0)
        assert not eng.variance_is_uniform()
        eng.set_data(case["data"], np.full(shape, 0.0))       # var == 0 -> 1e12 everywhere
        assert eng.variance_is_uniform()
        eng.upload_slot(_lib.SLOT_IVAR, 1.0 / case["var"])
        assert not eng.variance_is_uniform()


def test_residual_refresh_keeps_chain_consistent():
    """lib/run.py:521-534: the periodic from-scratch residual only removes
    ~1e-14 creep."""
    case = make_case("c1")
    with engine_for(case) as eng:
        eng.set_params(case["init"])
        eng.mh_config(case["min_b"], case["max_b"], 0.1, 50.0, seed=9, refresh_every=0)
        eng.mh_sweeps(5, 1)
        carried = eng.download_slot(2)
        fresh = eng.residual()
    scale = np.max(np.abs(fresh))
    assert np.max(np.abs(carried - fresh)) <= 1e-11 * scale


def test_proposal_table_is_rebuilt_for_every_call():
    """Option mh_props: the proposals of a sweep come from one launch before its colour
    launches.  The table never outlives a call: the same sweep number run twice in a row
    (from the state the first run left) and per-phase stepping give the same bits with the
    table as without it."""
    from deconv3d_amd import _lib
    case = make_case("moffat")
    outs = []
    for props in (1, 0):
        with engine_for(case, options={"mh_props": props}) as eng:
            eng.set_params(case["init"])
            eng.mh_config(case["min_b"], case["max_b"], 0.1, 50.0, seed=5, refresh_every=0)
            eng.mh_sweeps(1, 1)
            eng.mh_sweeps(1, 1)          # sweep number 1 again, other parameters now
            eng.mh_phase(0, 2)           # d3d_mh_phase: one call per phase
            eng.mh_sweeps(2, 3)
            outs.append((eng.get_params(), eng.download_slot(_lib.SLOT_ERR), eng.get_dlog()))
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("name,uniform", [("moffat", False), ("c1", False), ("odd_depth", False),
                                          ("c1", True), ("nolsf", False), ("big_fsf", False),
                                          ("rect_fsf", False), ("tiny", False), ("asym", False)])
def test_small_launch_kernel_is_bit_identical_to_the_round3_variants(name, uniform):
    """Round 4, k_mh_small (csrc/d3d_mh_small.h): colour launches that do not fill the chip --
    table-free window pass with batched LDS reads, the update's lines from the sweep's line
    table instead of a prepare wavefront, the decision on the wavefronts that hold channels --
    against k_mh_ws's small variants (option mh_small = 0): the same chain bit for bit
    (parameters, carried residual, log-ratio map, accepted count), across calls, a masked
    spaxel, per-phase stepping, a repeated sweep number and a from-scratch residual."""
    from deconv3d_amd import _lib
    case = make_case(name)
    outs = []
    for small in (1, 0):
        with engine_for(case, options={"mh_small": small}) as eng:
            if uniform:
                eng.set_data(case["data"], None, var_scalar=float(case["var"].mean()), mask=case["mask"])
                assert eng.variance_is_uniform()
            eng.set_params(case["init"])
            eng.mh_config(case["min_b"], case["max_b"], 0.1, 50.0, seed=5, refresh_every=3)
            acc = eng.mh_sweeps(2, 1)
            acc += eng.mh_sweeps(1, 1)          # sweep number 1 again, other parameters now
            eng.mh_phase(0, 2)
            acc += eng.mh_sweeps(3, 3)
            outs.append((eng.get_params(), eng.download_slot(_lib.SLOT_ERR), eng.get_dlog(),
                         np.array([acc])))
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)


def test_small_launch_kernel_wide_form_is_bit_identical_to_round3s():
    """The same for the WIDE form (704 streaming threads: the small launches of a partitioned
    128-channel context): a 128 x 30 x 40 cube cut into a far and a near
    part like a row-strip tile, zig-zag walk, masked spaxels."""
    from deconv3d_amd import _lib
    D, H, W = 128, 30, 40
    fsf = O.moffat_cropped(11, 3.0, 2.5)
    lsf = O.muse_like_lsf(D)
    data, var, mask, truth, init, mn, mx = O.synthetic_case(D, H, W, fsf, lsf, seed=77)
    mask[3, 5] = mask[20, 33] = 0
    outs = []
    for small in (1, 0):
        with _lib.Engine((D, H, W), fsf.shape, options={"mh_small": small}) as eng:
            eng.set_taps(fsf, lsf)
            eng.set_data(data, var, mask=mask)
            eng.set_parts([(0, H - 10, 0, W), (H - 10, H, 0, W)], [0, 1])
            eng.set_params(init)
            eng.mh_config(mn, mx, 0.1, float(mx[0] ** 2), seed=9, refresh_every=2)
            acc = eng.mh_sweeps(3, 1)
            outs.append((eng.get_params(), eng.download_slot(_lib.SLOT_ERR), eng.get_dlog(),
                         np.array([acc])))
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("kind", ["two layers", "one layer", "uniform variance", "beyond-cache policy",
                                  "two layers, 64 channels", "masked, odd depth"])
def test_table_kernel_chip_filling_form_is_bit_identical_to_k_mh_ws(kind):
    """k_mh_small's chip-filling form (option mh_small = 2: relative position tables, the sweep's
    line table, one or two pending layers, ONE deciding wavefront, optionally the non-temporal /
    write-through policy) against k_mh_ws on launches that fill the chip (>= 512 windows per
    colour class): the same chain bit for bit -- parameters, carried residual, log-ratio map,
    accepted count -- over three sweeps and a from-scratch residual.  (An EXPERIMENTS-build
    kernel: correct, and measured slower than k_mh_ws -- a launch that fills the chip is bound by
    HBM, not by instruction issue)"""
    from deconv3d_amd import _lib
    if not _lib.has_experiments():
        pytest.skip("the chip-filling form of k_mh_small is compiled with make EXPERIMENTS=1 only")
    rng = np.random.default_rng(31)
    D = {"two layers, 64 channels": 64, "masked, odd depth": 21}.get(kind, 16)
    H, W = 256, 254                                # 25 x 25 = 625 windows per colour class
    fsf = O.moffat_cropped(11, 3.0, 2.5)
    lsf = O.gaussian_lsf_vector(D, 0.9088)
    y, x = np.indices((H, W))
    truth = np.dstack((1.0 + 9.0 * np.exp(-((y - H / 2.) ** 2 + (x - W / 2.) ** 2) / (2. * 60. ** 2)),
                       D * (0.3 + 0.4 * rng.random((H, W))), 0.8 + 1.5 * rng.random((H, W))))
    mask = np.ones((H, W))
    if kind == "masked, odd depth":
        mask[rng.integers(0, H, 200), rng.integers(0, W, 200)] = 0
    opts = {"mh_layers": 1} if kind == "one layer" else {}
    if kind == "beyond-cache policy":
        opts["mh_nt_ivar"] = 1
    outs = []
    for small in (2, 1):
        with _lib.Engine((D, H, W), fsf.shape, options=dict(opts, mh_small=small)) as eng:
            eng.set_taps(fsf, lsf)
            eng.set_params(truth)
            clean = eng.forward()
            sigma = 0.05 * clean.max()
            data = clean + np.random.default_rng(5).normal(0., sigma, clean.shape)
            var = (sigma * (0.5 + np.random.default_rng(6).random(clean.shape))) ** 2
            if kind == "uniform variance":
                eng.set_data(data, None, var_scalar=sigma ** 2, mask=mask)
                assert eng.variance_is_uniform()
            else:
                eng.set_data(data, var, mask=mask)
            assert eng.mh_layers() == (1 if kind == "one layer" else 2)
            assert eng.get_option("small_parts") == (1 if small == 2 else 0)
            max_b = np.array([data.max() / fsf.max(), D - 1., float(D)])
            init = max_b * np.random.default_rng(7).random((H, W, 3))
            init[..., 2] = np.maximum(init[..., 2], 0.3)
            eng.set_params(init)
            eng.mh_config(np.zeros(3), max_b, 0.1, float(max_b[0] ** 2), seed=11, refresh_every=2)
            acc = eng.mh_sweeps(3, 1)
            outs.append((eng.get_params(), eng.download_slot(_lib.SLOT_ERR), eng.get_dlog(), np.array([acc])))
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)


def test_options_belong_to_a_context_not_to_the_process(monkeypatch):
    """VERDICT r2 item 8 on the default build: three contexts of one process with different
    kernel families, alive together and stepped in turn, each keep their own setting and
    produce the same chain bit for bit; an option can change between calls (pending
    layers are flushed); the environment only supplies the default of a NEW context."""
    from deconv3d_amd import _lib
    case = make_case("c1")
    engs = []
    for opts in ({"mh_layers": 2}, {"mh_layers": 3}, {"mh_defer": 0}):
        e = engine_for(case, options=opts)
        e.set_params(case["init"])
        e.mh_config(case["min_b"], case["max_b"], 0.1, 50.0, seed=5, refresh_every=0)
        engs.append(e)
    try:
        assert [e.get_option("mh_defer") for e in engs] == [1, 1, 0]
        assert [e.mh_layers() for e in engs][:2] == [2, 3]
        for s in (1, 2, 3):
            for e in engs:
                e.mh_sweeps(1, s)
        engs[1].set_option("mh_layers", 1)           # mid-chain: flushes what is pending
        assert engs[1].mh_layers() == 1 and engs[0].mh_layers() == 2
        for e in engs:
            e.mh_sweeps(1, 4)
        outs = [(e.get_params(), e.download_slot(_lib.SLOT_ERR)) for e in engs]
        for other in outs[1:]:
            np.testing.assert_array_equal(other[0], outs[0][0])
            np.testing.assert_array_equal(other[1], outs[0][1])
        with pytest.raises(ValueError):
            engs[0].set_option("no_such_option", 1)
        with pytest.raises(ValueError):
            engs[0].set_option("mh_layers", 7)
        # the environment: a default for contexts created afterwards, nothing more
        monkeypatch.setenv("D3D_MH_DEFER", "2")
        assert engs[0].get_option("mh_defer") == 1
        with engine_for(case) as late:
            assert late.get_option("mh_defer") == 2
        with engine_for(case, options={"mh_defer": 0}) as late:
            assert late.get_option("mh_defer") == 0
    finally:
        for e in engs:
            e.close()
