"""
GPU tests of the drop-in Run() API and of the committed golden vectors
(tests/golden/*.npz: produced in the build container by make_goldens.py; the
reference itself is not available on the GPU box).
"""
import os

import numpy as np
import pytest

import deconv3d_amd as d3d
from deconv3d_amd import _lib

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name), allow_pickle=False)


@pytest.mark.parametrize("name", ["c1", "odd_depth", "asym"])
def test_device_matches_committed_golden_vectors(name):
    g = gold("oracle_%s.npz" % name)
    shape = g["data"].shape
    with _lib.Engine(shape, g["fsf"].shape) as eng:
        eng.set_taps(g["fsf"], g["lsf"])
        eng.set_data(g["data"], g["var"], mask=g["mask"])
        eng.set_params(g["truth"])
        tol = 1e-12
        assert np.max(np.abs(eng.build_clean() - g["clean"])) <= tol * np.max(np.abs(g["clean"]))
        assert np.max(np.abs(eng.forward() - g["sim"])) <= tol * np.max(np.abs(g["sim"]))
        eng.set_params(g["init"])
        assert np.max(np.abs(eng.residual() - g["err"])) <= tol * np.max(np.abs(g["err"]))
        cmap, total = eng.chi2_map()
        np.testing.assert_allclose(cmap, g["chi2_map"], rtol=1e-10,
                                   atol=1e-12 * g["chi2_map"].sum())
        for rec in g["probes"]:
            y, x = int(rec[0]), int(rec[1])
            got = eng.window_stats(y, x, rec[2:5])
            np.testing.assert_allclose(got[:3], rec[5:8], rtol=1e-10,
                                       atol=1e-12 * max(rec[5], rec[6]))
            np.testing.assert_allclose(got[3:], rec[8:], rtol=1e-10,
                                       atol=1e-12 * max(abs(rec[8]), abs(rec[9])))
        # deterministic chain: same Philox stream and colour order as the oracle
        eng.mh_config(g["min_b"], g["max_b"], 0.1, float(g["ra"]), seed=int(g["chain_seed"]),
                      refresh_every=0)
        n = g["chain"].shape[0] - 1
        chain = np.full((n + 1,) + g["init"].shape, np.nan)
        dlog = np.full((n + 1,) + g["mask"].shape, np.nan)
        acc = eng.mh_sweeps(n, 1, 1, chain, dlog)
        live = g["mask"] == 1
        for s in range(1, n + 1):
            np.testing.assert_allclose(chain[s][live], g["chain"][s][live], rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(dlog[s][live], g["dlog"][s - 1][live], rtol=1e-8,
                                       atol=1e-10 * (np.max(np.abs(g["dlog"][s - 1][live])) + 1))
        assert acc == int(g["accepted"])


def test_reference_mat_fixture_on_device():
    """The reference's own data set (tests/input/data14forAntoine.mat): with
    the theoretical parameters the normalised residual of the DEVICE forward
    model is white (std 1.000), and a short chain started there stays there."""
    g = gold("ref_mat_fixture.npz")
    data, var, fsf, params = g["data"], g["var"], g["fsf"], g["params"]
    D, H, W = data.shape
    with _lib.Engine((D, H, W), fsf.shape) as eng:
        eng.set_taps(fsf, None)
        eng.set_data(data, var)
        eng.set_params(params)
        err = eng.residual()
        z = err / np.sqrt(var)
        assert abs(z.mean()) < 0.02 and abs(z.std() - 1.0) < 0.02
        cmap, total = eng.chi2_map()
        assert abs(2 * total / data.size - 1.0) < 0.04


def synthetic_cube(D=32, H=16, W=16, seed=12345, A0=10.0):
    """BASELINE config 1: Gaussian FSF FWHM 3 px, Gaussian LSF
    FWHM 2.675 A, built through the product's own plugin classes + device
    forward model."""
    inst = d3d.MUSE(fsf_fwhm=0.6)
    rng = np.random.default_rng(seed)
    y, x = np.indices((H, W))
    r2 = (y - H / 2.) ** 2 + (x - W / 2.) ** 2
    truth = np.dstack((A0 * np.exp(-r2 / (2. * (H / 6.) ** 2)),
                       D / 2. + (D / 8.) * np.tanh((x - W / 2.) / (W / 8.)),
                       rng.uniform(1.5, 3.0, size=(H, W))))
    blank = inst.build_cube(np.ones((D, H, W)))
    fsf, lsf = inst.fsf.as_image(blank), inst.lsf.as_vector(blank)
    with _lib.Engine((D, H, W), fsf.shape) as eng:
        eng.set_taps(fsf, lsf)
        eng.set_params(truth)
        clean = eng.forward()
    sigma = 0.05 * A0 * fsf.max()
    data = clean + rng.normal(0., sigma, size=clean.shape)
    return inst, inst.build_cube(data), np.full(clean.shape, sigma ** 2), truth, sigma


def test_run_end_to_end_config1():
    """BASELINE config 1: 32x16x16, GaussianFieldSpreadFunction FWHM 3 px,
    Gaussian LSF, SingleGaussianLineModel, 200 iterations through Run()."""
    inst, cube, var, truth, sigma = synthetic_cube()
    run = d3d.Run(cube, inst, variance=var, max_iterations=200, keep_one_in=2,
                  jump_amplitude=[0., 0.5, 0.2], seed=1)
    assert run.chain.shape == (100, 16, 16, 3) and run.likelihoods.shape == (100, 16, 16)
    assert run.fsf.shape == (9, 9) and run.lsf.shape == (32,)
    assert run.parameters.shape == (16, 16, 3)
    assert run.convolved_cube.data.shape == cube.data.shape
    assert run.clean_cube.data.shape == cube.data.shape
    assert not np.isnan(run.chain).any()                     # every slot written
    assert 0.0 < run.acceptance_rate <= 1.0
    # the fit explains the data: reduced chi2 of the last sweep close to 1
    run.engine.set_params(run.chain[-1])
    err = run.engine.residual()
    red = np.sum(err ** 2 / var) / err.size
    assert red < 2.0, red
    # bright spaxels recover their line centre to within the (3 px FWHM) blur
    bright = truth[..., 0] > 5.0
    assert np.median(np.abs(run.parameters[..., 1] - truth[..., 1])[bright]) < 1.5
    # bounds respected (lib/run.py:379-388)
    assert (run.chain[..., 0] >= 0).all() and (run.chain[..., 0] <= run.max_boundaries[0]).all()
    assert (run.chain[..., 2] >= 0).all() and (run.chain[..., 2] <= 32).all()


def test_run_end_to_end_on_a_deep_cube_with_the_muse_lsf():
    """Run() on a 600-channel cube with MUSELineSpreadFunction's analytic stand-in (taps
    within +-8 channels): the z-blocked sweep kernels, the blocked LSF pass and the z-blocked
    FSF pass behind the reference's API -- streamed chain, fit quality, bounds."""
    D, H, W = 600, 12, 11
    inst = d3d.MUSE(lsf=d3d.MUSELineSpreadFunction(model="analytic"), fsf_fwhm=0.6)
    rng = np.random.default_rng(5)
    y, x = np.indices((H, W))
    r2 = (y - H / 2.) ** 2 + (x - W / 2.) ** 2
    truth = np.dstack((10.0 * np.exp(-r2 / (2. * (H / 5.) ** 2)),
                       D / 2. + (D / 10.) * np.tanh((x - W / 2.) / (W / 6.)),
                       rng.uniform(1.5, 3.0, size=(H, W))))
    blank = inst.build_cube(np.ones((D, H, W)))
    fsf, lsf = inst.fsf.as_image(blank), inst.lsf.as_vector(blank)
    with _lib.Engine((D, H, W), fsf.shape) as eng:
        eng.set_taps(fsf, lsf)
        eng.set_params(truth)
        clean = eng.forward()
        assert eng.get_option("mh_zblocks") == 1
    sigma = 0.05 * 10.0 * fsf.max()
    data = clean + rng.normal(0., sigma, size=clean.shape)
    var = np.full(clean.shape, sigma ** 2) * (0.5 + rng.random(clean.shape))
    # (a 0.5-channel random walk does not find a line among 600 channels in a test's time:
    # start near it, as a user of a deep cube would -- initial_parameters, lib/run.py:290-312)
    start = truth + np.dstack((rng.normal(0., 1., (H, W)), rng.normal(0., 3., (H, W)),
                               rng.normal(0., 0.3, (H, W))))
    start[..., 0] = np.abs(start[..., 0])
    run = d3d.Run(inst.build_cube(data), inst, variance=var, max_iterations=120, keep_one_in=3,
                  jump_amplitude=[0., 0.5, 0.2], initial_parameters=start, seed=2)
    assert run.chain.shape == (40, H, W, 3) and not np.isnan(run.chain).any()
    assert 0.0 < run.acceptance_rate <= 1.0
    run.engine.set_params(run.chain[-1])
    err = run.engine.residual()
    assert np.sum(err ** 2 / var) / err.size < 2.0
    bright = truth[..., 0] > 5.0
    assert np.median(np.abs(run.parameters[..., 1] - truth[..., 1])[bright]) < 2.0
    assert (run.chain[..., 0] >= 0).all() and (run.chain[..., 2] <= D).all()


def test_concurrent_chains_on_one_gpu_are_the_chains_they_would_be_alone():
    """deconv3d_amd.ensemble.sweep_chains: four contexts of one process, one host thread
    each, their colour launches overlapping on the device -- every chain, its streamed
    samples and its accepted count equal the same context run alone, bit for bit."""
    from deconv3d_amd import ensemble
    from tests.cases import make_case
    case = make_case("c1")
    D, H, W = case["D"], case["H"], case["W"]

    def make(seed):
        eng = _lib.Engine((D, H, W), case["fsf"].shape)
        eng.set_taps(case["fsf"], case["lsf"])
        eng.set_data(case["data"], case["var"], mask=case["mask"])
        eng.set_params(case["init"])
        eng.mh_config(case["min_b"], case["max_b"], 0.1, 40.0, seed=seed, refresh_every=0)
        return eng

    n = 12
    alone = []
    for seed in (1, 2, 3, 4):
        with make(seed) as eng:
            chain = np.full((n + 1, H, W, 3), np.nan)
            acc = eng.mh_sweeps(n, 1, 1, chain, None)
            alone.append((chain, acc, eng.download_slot(_lib.SLOT_ERR)))
    engs = [make(seed) for seed in (1, 2, 3, 4)]
    try:
        chains = [np.full((n + 1, H, W, 3), np.nan) for _ in engs]
        accs = ensemble.sweep_chains(engs, n, 1, 1, chains)
        for i, eng in enumerate(engs):
            np.testing.assert_array_equal(chains[i][1:], alone[i][0][1:])
            assert accs[i] == alone[i][1]
            np.testing.assert_array_equal(eng.download_slot(_lib.SLOT_ERR), alone[i][2])
        assert not np.array_equal(chains[0][-1], chains[1][-1])
    finally:
        for e in engs:
            e.close()


@pytest.mark.parametrize("name,n_chains", [("c1", 5), ("tile_a", 16), ("odd_depth", 3)])
def test_batched_chains_are_the_chains_they_would_be_alone(name, n_chains):
    """d3d_mh_sweeps_batch (ensemble.sweep_chains_batched): R contexts of one geometry -- other
    data, bounds, start and seed each -- in ONE launch per colour class: parameters, carried
    residual, log-ratio map and accepted count of every chain equal those of its context run
    alone, bit for bit (whatever pending-layer depth the joint launch takes), across two calls
    and a periodic residual rebuild; contexts that do not share mask or taps are refused."""
    from deconv3d_amd import ensemble
    from tests.cases import make_case
    case = make_case(name)     # (tile_a x 16: the joint launch fills the chip -- two pending layers)
    D, H, W = case["D"], case["H"], case["W"]
    rng = np.random.default_rng(17)

    def make(r):
        eng = _lib.Engine((D, H, W), case["fsf"].shape)
        eng.set_taps(case["fsf"], case["lsf"])
        data = case["data"] * (1.0 + 0.1 * r) + 0.01 * r
        eng.set_data(data, case["var"] * (1.0 + 0.05 * r), mask=case["mask"])
        init = case["init"].copy()
        init[..., 1] = np.clip(init[..., 1] + 0.3 * r, 0, D - 1)
        eng.set_params(init)
        max_b = case["max_b"] * np.array([1.0 + 0.1 * r, 1.0, 1.0])
        eng.mh_config(case["min_b"], max_b, 0.1, 40.0 + r, seed=100 + r, refresh_every=4)
        return eng

    alone = []
    for r in range(n_chains):
        with make(r) as eng:
            chain = np.full((4, H, W, 3), np.nan)
            dlog = np.full((4, H, W), np.nan)
            acc = eng.mh_sweeps(3, 1, 2, chain, dlog) + eng.mh_sweeps(4, 4, 2, chain, dlog)
            alone.append((eng.get_params(), eng.download_slot(_lib.SLOT_ERR), eng.get_dlog(), acc, chain, dlog))
    engs = [make(r) for r in range(n_chains)]
    try:
        # (saved sweeps streamed per chain, every second sweep; the last engine opts out)
        chains = [np.full((4, H, W, 3), np.nan) for _ in engs[:-1]] + [None]
        dlogs = [np.full((4, H, W), np.nan) for _ in engs[:-1]] + [None]
        a1 = ensemble.sweep_chains_batched(engs, 3, 1, 2, chains, dlogs)
        a2 = ensemble.sweep_chains_batched(engs, 4, 4, 2, chains, dlogs)
        for r, eng in enumerate(engs):
            np.testing.assert_array_equal(eng.get_params(), alone[r][0])
            np.testing.assert_array_equal(eng.download_slot(_lib.SLOT_ERR), alone[r][1])
            np.testing.assert_array_equal(eng.get_dlog(), alone[r][2])
            assert a1[r] + a2[r] == alone[r][3]
            if chains[r] is not None:
                np.testing.assert_array_equal(chains[r][1:], alone[r][4][1:])     # slots 1..3: sweeps 2, 4, 6
                np.testing.assert_array_equal(dlogs[r][1:], alone[r][5][1:])
                assert np.isnan(chains[r][0]).all()
        assert not np.array_equal(alone[0][0], alone[1][0])
        # a chain goes on alone afterwards
        engs[1].mh_sweeps(1, 8)
        other_mask = case["mask"].copy()
        other_mask[0, 0] = 1 - other_mask[0, 0]
        engs[0].set_data(case["data"], case["var"], mask=other_mask)
        with pytest.raises(ValueError, match="mask"):
            ensemble.sweep_chains_batched(engs, 1, 9)
    finally:
        for e in engs:
            e.close()


def test_batched_chains_then_alone_whatever_layers_were_left_pending():
    """a joint launch that fills the chip keeps TWO pending layers, and after an
    even number of colour launches since the last flush both are still pending when the call
    returns (tile_a: 35 colour classes; 2 sweeps = 70 launches).  A context that then runs
    ALONE -- one layer for its small launches -- must apply them instead of failing, and stay
    the chain it would have been alone; so must it after an odd count (3 sweeps).  Contexts
    that rebuild their residual at different sweeps cannot share the pending layers: refused."""
    from deconv3d_amd import ensemble
    from tests.cases import make_case
    case = make_case("tile_a")
    D, H, W = case["D"], case["H"], case["W"]

    def make(r, refresh_every=0):
        eng = _lib.Engine((D, H, W), case["fsf"].shape)
        eng.set_taps(case["fsf"], case["lsf"])
        eng.set_data(case["data"] * (1.0 + 0.1 * r), case["var"], mask=case["mask"])
        eng.set_params(case["init"])
        eng.mh_config(case["min_b"], case["max_b"], 0.1, 40.0 + r, seed=300 + r,
                      refresh_every=refresh_every)
        return eng

    for n_batched in (2, 3):
        with make(1) as eng:
            eng.mh_sweeps(n_batched + 2, 1)
            want = (eng.get_params(), eng.download_slot(_lib.SLOT_ERR))
        engs = [make(r) for r in range(16)]
        try:
            ensemble.sweep_chains_batched(engs, n_batched, 1)
            engs[1].mh_sweeps(2, n_batched + 1)
            np.testing.assert_array_equal(engs[1].get_params(), want[0])
            np.testing.assert_array_equal(engs[1].download_slot(_lib.SLOT_ERR), want[1])
        finally:
            for e in engs:
                e.close()
    engs = [make(0), make(1, refresh_every=5)]
    try:
        with pytest.raises(ValueError, match="refresh_every"):
            ensemble.sweep_chains_batched(engs, 1, 1)
    finally:
        for e in engs:
            e.close()


def test_run_is_reproducible_and_seed_sensitive():
    inst, cube, var, _, _ = synthetic_cube(D=16, H=9, W=9, seed=3)
    a = d3d.Run(cube, inst, variance=var, max_iterations=6, seed=11)
    b = d3d.Run(cube, inst, variance=var, max_iterations=6, seed=11)
    c = d3d.Run(cube, inst, variance=var, max_iterations=6, seed=12)
    np.testing.assert_array_equal(a.chain, b.chain)
    assert not np.array_equal(a.chain, c.chain)


@pytest.mark.parametrize("batched", [True, False])
def test_run_chains_are_the_single_runs_of_their_seeds(batched, tmp_path):
    """Run(..., chains=R) (round 4): R independent chains of one cube advanced together --
    one launch per colour class for all of them (d3d_mh_sweeps_batch), or, where the library
    cannot batch, concurrently on their own streams.  Chain r IS the chain of
    Run(..., seed=seed + r): same start, same samples, same log ratios, bit for bit;
    extract_parameters pools the chains; rhat is the per-parameter Gelman-Rubin map."""
    inst, cube, var, _, _ = synthetic_cube(D=16, H=9, W=9, seed=3)
    kw = dict(variance=var, max_iterations=13, keep_one_in=2, min_acceptance_rate=0.)

    class Threads(d3d.Run):
        _batched = False                      # (forces ensemble.sweep_chains)

    cls = d3d.Run if batched else Threads
    multi = cls(cube, inst, seed=11, chains=3, chain_file=str(tmp_path / "m") if batched else None, **kw)
    assert multi.n_chains == 3 and len(multi.chains) == 3 and multi.chain is multi.chains[0]
    means = []
    for r in range(3):
        one = d3d.Run(cube, inst, seed=11 + r, **kw)
        np.testing.assert_array_equal(multi.chains[r], one.chain)
        np.testing.assert_array_equal(multi.all_likelihoods[r][1:], one.likelihoods[1:])
        assert multi.acceptance_rates[r] == one.acceptance_rate
        means.append(one.extract_parameters())
    np.testing.assert_allclose(multi.parameters, np.mean(means, axis=0), rtol=1e-14)
    assert not np.array_equal(multi.chains[0][-1], multi.chains[1][-1])
    assert multi.rhat.shape == (9, 9, 3) and np.all(multi.rhat[np.isfinite(multi.rhat)] > 0.5)
    assert abs(multi.acceptance_rate - np.mean(multi.acceptance_rates)) < 1e-12
    if batched:                               # chain r > 0 of a memory-mapped run: <prefix>_c<r>_*
        np.testing.assert_array_equal(np.load(str(tmp_path / "m_c2_chain.npy")), multi.chains[2])
    # one start map per chain
    starts = np.stack([multi.chains[r][3] for r in range(3)])
    again = d3d.Run(cube, inst, seed=5, chains=3, initial_parameters=starts, **kw)
    for r in range(3):
        np.testing.assert_array_equal(again.chains[r][0], starts[r])
    with pytest.raises(ValueError, match="one map per chain"):
        d3d.Run(cube, inst, seed=5, chains=2, initial_parameters=starts, **kw)


def test_chains_checkpoint_and_resume(tmp_path):
    """checkpoint= / resume_state= with chains=R: the checkpoint holds the R parameter maps
    (what initial_parameters= takes for R chains), every chain's slots and accepted count;
    6 + 6 sweeps of three chains equal their 12 sweeps in one go, acceptance rates included."""
    inst, cube, var, _, _ = synthetic_cube(D=16, H=9, W=9, seed=4)
    name = str(tmp_path / "ck")
    kw = dict(variance=var, seed=3, chains=3, min_acceptance_rate=0., refresh_every=0)
    whole = d3d.Run(cube, inst, max_iterations=13, **kw)
    first = d3d.Run(cube, inst, max_iterations=7, write_every=7, checkpoint=name, **kw)
    state = np.load(name + "_state.npz")
    assert int(state["iteration"]) == 7 and int(state["n_chains"]) == 3
    assert int(state["per_chain_accepted"].sum()) == int(state["total_accepted"])
    saved = np.load(name + "_parameters.npy")
    assert saved.shape == (3, 9, 9, 3)
    for r in range(3):
        np.testing.assert_array_equal(first.chains[r][-1], saved[r])
        np.testing.assert_array_equal(
            np.load(name + ("_chain.npy" if r == 0 else "_c%d_chain.npy" % r)), first.chains[r])
    second = d3d.Run(cube, inst, max_iterations=7, initial_parameters=name + "_parameters.npy",
                     resume_state=name + "_state.npz", **kw)
    for r in range(3):
        # (the resumed run rebuilds the residual from the parameters: rounding-level differences)
        np.testing.assert_allclose(second.chains[r][-1], whole.chains[r][-1], rtol=1e-8, atol=1e-8)
        assert abs(second.acceptance_rates[r] - whole.acceptance_rates[r]) < 1e-12
    assert abs(second.acceptance_rate - whole.acceptance_rate) < 1e-12
    with pytest.raises(ValueError, match="holds 3 chain"):
        d3d.Run(cube, inst, max_iterations=3, resume_state=name + "_state.npz", variance=var)


def test_run_initial_parameters_and_mask(tmp_path):
    """lib/run.py:294-307 (3-D, 1-D broadcast, .npy path) and masks: masked
    spaxels keep their parameters and contribute nothing (lib/run.py:553-566)."""
    inst, cube, var, truth, _ = synthetic_cube(D=16, H=9, W=9, seed=5)
    p1 = np.array([2.0, 8.0, 1.5])
    run = d3d.Run(cube, inst, variance=var, initial_parameters=p1, max_iterations=1)
    np.testing.assert_array_equal(run.extract_parameters(), np.tile(p1, (9, 9, 1)))
    p3 = np.tile(p1, (9, 9, 1)) * np.linspace(0.5, 1.5, 81).reshape(9, 9, 1)
    run = d3d.Run(cube, inst, variance=var, initial_parameters=p3, max_iterations=1)
    np.testing.assert_array_equal(run.extract_parameters(), p3)
    path = str(tmp_path / "p.npy")
    np.save(path, p3)
    run = d3d.Run(cube, inst, variance=var, initial_parameters=path, max_iterations=1)
    np.testing.assert_array_equal(run.extract_parameters(), p3)
    mask = d3d.above_percentile(cube, 60)
    user_mask = mask.copy()
    run = d3d.Run(cube, inst, variance=var, mask=mask, initial_parameters=p3, max_iterations=5)
    np.testing.assert_array_equal(mask, user_mask)            # not mutated
    dead = mask == 0
    np.testing.assert_array_equal(run.chain[-1][dead], p3[dead])
    assert not np.array_equal(run.chain[-1][~dead], p3[~dead])
    assert np.all(run.clean_cube.data[:, dead] == 0)


def test_run_default_variance_and_save(tmp_path):
    """variance=None -> median-clipped sigma of the reference's sub-block
    (lib/run.py:186-192); save() writes the reference's files (lib/run.py:742-788)."""
    import matplotlib
    matplotlib.use("Agg")
    inst, cube, var, _, sigma = synthetic_cube(D=24, H=12, W=12, seed=8)
    run = d3d.Run(cube, inst, max_iterations=4)
    est = np.sqrt(run.variance_cube.flat[0])
    assert np.all(run.variance_cube == run.variance_cube.flat[0]) and est > 0
    name = str(tmp_path / "out")
    run.save(name, clobber=True)
    for suffix in ("_parameters.npy", "_chain.npy", "_matlab.mat", "_images.png", "_chain.png",
                   "_convolved_cube.fits", "_clean_cube.fits", "_result.npz"):
        assert os.path.isfile(name + suffix), suffix
    back = d3d.Cube.from_fits(name + "_convolved_cube.fits")
    np.testing.assert_array_equal(back.data, run.convolved_cube.data)
    # resume from the saved parameters (lib/run.py:790-797 -> :294-307)
    again = d3d.Run(cube, inst, initial_parameters=name + "_parameters.npy", max_iterations=1)
    np.testing.assert_array_equal(again.extract_parameters(), run.extract_parameters())


def test_contribution_of_spaxel_matches_forward():
    inst, cube, var, truth, _ = synthetic_cube(D=16, H=9, W=9, seed=6)
    run = d3d.Run(cube, inst, variance=var, max_iterations=1)
    total = np.zeros_like(cube.data)
    for (y, x) in [(0, 0), (4, 4), (8, 3)]:
        c, _ = run.contribution_of_spaxel(x, y, truth[y, x], 9, 9, 16)
        assert c.shape == cube.data.shape
        fh = (run.fsf.shape[0] - 1) // 2
        out = np.ones((9, 9), bool)
        out[max(0, y - fh):y + fh + 1, max(0, x - fh):x + fh + 1] = False
        assert np.all(c[:, out] == 0)                        # only the FSF window is touched
        total += c
    assert total.max() > 0


def test_min_acceptance_rate_stops_the_loop():
    """lib/run.py:344-350: the loop ends once the running acceptance falls
    to min_acceptance_rate."""
    inst, cube, var, _, _ = synthetic_cube(D=16, H=9, W=9, seed=7)
    run = d3d.Run(cube, inst, variance=var, max_iterations=50, min_acceptance_rate=0.999,
                  jump_amplitude=5.0)
    assert run.iterations_done < 50


def test_write_every_checkpoints_and_resume(tmp_path):
    """`write_every` (accepted but unused by the reference, lib/run.py:89-92,107)
    is the checkpoint cadence: the parameter map on disk restarts a run."""
    inst, cube, var, _, _ = synthetic_cube(D=16, H=9, W=9, seed=4)
    name = str(tmp_path / "ck")
    run = d3d.Run(cube, inst, variance=var, max_iterations=13, write_every=4, keep_one_in=2,
                  checkpoint=name, seed=3)
    ck = np.load(name + "_parameters.npy")
    assert ck.shape == (9, 9, 3)
    chain = np.load(name + "_chain.npy")
    np.testing.assert_array_equal(chain, run.chain[:chain.shape[0]])
    again = d3d.Run(cube, inst, variance=var, initial_parameters=name + "_parameters.npy",
                    max_iterations=1)
    np.testing.assert_array_equal(again.chain[0], ck)


def test_contribution_of_spaxel_values_match_the_oracle():
    """Run.contribution_of_spaxel (lib/run.py:654-708) value by value against the
    oracle's restatement: a corner, an edge, an interior and a MASKED spaxel (the
    reference ignores the mask here), and the device chain state is untouched."""
    from oracle import deconv3d_oracle as O
    inst, cube, var, truth, _ = synthetic_cube(D=16, H=9, W=9, seed=6)
    mask = np.ones((9, 9))
    mask[5, 2] = 0
    run = d3d.Run(cube, inst, variance=var, mask=mask, max_iterations=3, seed=2)
    before = run.engine.get_params()
    for (y, x) in [(0, 0), (8, 8), (0, 4), (4, 4), (5, 2)]:
        p = truth[y, x] * np.array([1.3, 1.0, 0.8])
        got, _ = run.contribution_of_spaxel(x, y, p, 9, 9, 16)
        want = O.contribution_of_spaxel(x, y, p, 9, 9, 16, run.fsf, run.lsf)
        assert np.max(np.abs(got - want)) <= 1e-12 * np.max(np.abs(want)), (y, x)
    np.testing.assert_array_equal(run.engine.get_params(), before)
    # simulate_* take explicit parameters and leave the chain state alone as well
    sim = run.simulate_convolved(cube.data.shape, truth)
    want = O.forward_full(cube.data.shape, truth, mask, run.fsf, run.lsf)
    assert np.max(np.abs(sim - want)) <= 1e-12 * np.max(np.abs(want))
    np.testing.assert_array_equal(run.engine.get_params(), before)


def test_run_on_a_cube_with_nan_voxels():
    """A cube with one NaN voxel and one all-NaN spaxel goes through Run():
    NaN spectra are masked (lib/run.py:157-162), the bounds stay finite
    (lib/line_models.py:79-90 would make them NaN) and the chain moves."""
    inst, cube, var, truth, _ = synthetic_cube(D=16, H=9, W=9, seed=9)
    data = cube.data.copy()
    data[3, 2, 6] = np.nan          # one voxel
    data[:, 7, 1] = np.nan          # a whole spectrum
    bad = inst.build_cube(data)
    run = d3d.Run(bad, inst, variance=var, max_iterations=6, seed=5)
    assert np.all(np.isfinite(run.max_boundaries)) and run.max_boundaries[0] > 0
    assert run.mask[2, 6] == 0 and run.mask[7, 1] == 0 and run.mask.sum() == 79
    live = run.mask == 1
    assert np.all(np.isfinite(run.chain[:, live]))
    assert not np.array_equal(run.chain[-1][live], run.chain[0][live])
    np.testing.assert_array_equal(run.chain[-1][~live], run.chain[0][~live])
    assert np.all(np.isfinite(run.convolved_cube.data))
    # default variance on a NaN cube: median_clip ignores the NaNs
    run2 = d3d.Run(bad, inst, max_iterations=2, seed=5)
    assert np.isfinite(run2.variance_cube).all()


def test_resume_continues_the_random_streams(tmp_path):
    """A run resumed from a checkpoint (parameters + state) continues the sweep
    numbering: 6 + 6 sweeps equal 12 sweeps in one go, and without the state the
    second segment would replay the first segment's random numbers."""
    inst, cube, var, _, _ = synthetic_cube(D=16, H=9, W=9, seed=4)
    name = str(tmp_path / "ck")
    kw = dict(variance=var, seed=3, min_acceptance_rate=0., refresh_every=0)
    whole = d3d.Run(cube, inst, max_iterations=13, **kw)
    first = d3d.Run(cube, inst, max_iterations=7, write_every=7, checkpoint=name, **kw)
    state = np.load(name + "_state.npz")
    assert int(state["iteration"]) == 7 and int(state["seed"]) == 3
    second = d3d.Run(cube, inst, max_iterations=7, initial_parameters=name + "_parameters.npy",
                     resume_state=name + "_state.npz", **kw)
    np.testing.assert_array_equal(first.chain[-1], np.load(name + "_parameters.npy"))
    # (the resumed run rebuilds the residual from the parameters: rounding-level differences)
    np.testing.assert_allclose(second.chain[-1], whole.chain[-1], rtol=1e-8, atol=1e-8)
    replay = d3d.Run(cube, inst, max_iterations=7, initial_parameters=name + "_parameters.npy",
                     **kw)
    assert not np.allclose(replay.chain[-1], whole.chain[-1], rtol=1e-3, atol=1e-3)


def test_resumed_run_carries_the_acceptance_count_and_memmapped_checkpoints_share_a_prefix(tmp_path):
    """(i) chain_file= and checkpoint= may name the same prefix -- the
    checkpoint flushes the memory-mapped chain instead of rewriting the file under the
    open mapping; (ii) the running acceptance rate of the stopping rule
    (lib/run.py:344-359) is that of the whole chain, earlier segments included."""
    inst, cube, var, _, _ = synthetic_cube(D=16, H=9, W=9, seed=4)
    name = str(tmp_path / "same")
    kw = dict(variance=var, seed=3, min_acceptance_rate=0., refresh_every=0)
    first = d3d.Run(cube, inst, max_iterations=9, write_every=2, keep_one_in=1, checkpoint=name,
                    chain_file=name, **kw)
    plain = d3d.Run(cube, inst, max_iterations=9, keep_one_in=1, **kw)
    np.testing.assert_array_equal(np.asarray(first.chain), plain.chain)
    np.testing.assert_array_equal(np.load(name + "_chain.npy", mmap_mode="r"), plain.chain)
    state = np.load(name + "_state.npz")
    assert str(state["chain_file"]) == name
    n_sp = 81
    second = d3d.Run(cube, inst, max_iterations=5, initial_parameters=name + "_parameters.npy",
                     resume_state=name + "_state.npz", **kw)
    # whole-chain rate: accepted of both segments over the iterations of both
    it_prev, acc_prev = int(state["total_iterations"]), int(state["total_accepted"])
    assert second._it_base == it_prev - 1 and second._acc_base == acc_prev - n_sp
    total_it = second.iterations_done + it_prev - 1
    assert 0. < second.acceptance_rate <= 1.
    assert abs(second.acceptance_rate * n_sp * total_it - round(second.acceptance_rate * n_sp * total_it)) < 1e-6


def test_streamed_chain_equals_the_synchronous_one():
    """d3d_mh_sweeps streams saved sweeps through device snapshots, a copy stream and
    pinned buffers (more saved sweeps than buffers, so they are recycled): the chain
    and the log ratios it delivers equal what stopping after every sweep and reading
    the state back gives (lib/run.py:428-432, 447-451)."""
    from oracle import deconv3d_oracle as O
    D, H, W = 16, 11, 13
    fsf, lsf = O.gaussian_fsf_image(2.0), O.gaussian_lsf_vector(D, 0.7)
    data, var, mask, truth, init, mn, mx = O.synthetic_case(D, H, W, fsf, lsf, seed=3)
    n = 23

    def engine():
        e = _lib.Engine((D, H, W), fsf.shape)
        e.set_taps(fsf, lsf)
        e.set_data(data, var, mask=mask)
        e.set_params(init)
        e.mh_config(mn, mx, 0.1, float(mx[0] ** 2), seed=5, refresh_every=7)
        return e
    for keep in (1, 3):
        slots = n // keep + 1
        chain = np.full((slots, H, W, 3), np.nan)
        dlog = np.full((slots, H, W), np.nan)
        with engine() as e:
            acc = e.mh_sweeps(n, 1, keep, chain, dlog)
        want_chain = np.full_like(chain, np.nan)
        want_dlog = np.full_like(dlog, np.nan)
        total = 0
        with engine() as e:
            for s in range(1, n + 1):
                total += e.mh_sweeps(1, s)
                if s % keep == 0:
                    want_chain[s // keep] = e.get_params()
                    want_dlog[s // keep] = e.get_dlog()
        np.testing.assert_array_equal(chain, want_chain)
        np.testing.assert_array_equal(dlog, want_dlog)
        assert acc == total


def test_memory_mapped_chain(tmp_path):
    """chain_file=: chain and likelihoods are written into memory-mapped .npy files as
    the device streams them out; same numbers as the in-memory chain, early stop
    leaves NaN slots, and the files reload with numpy."""
    inst, cube, var, _, _ = synthetic_cube(D=16, H=9, W=9, seed=4)
    kw = dict(variance=var, seed=3, keep_one_in=2, min_acceptance_rate=0.)
    ram = d3d.Run(cube, inst, max_iterations=13, **kw)
    name = str(tmp_path / "big")
    disk = d3d.Run(cube, inst, max_iterations=13, chain_file=name, **kw)
    assert isinstance(disk.chain, np.memmap)
    np.testing.assert_array_equal(np.asarray(disk.chain), ram.chain)
    np.testing.assert_array_equal(np.asarray(disk.likelihoods), ram.likelihoods)
    back = np.load(name + "_chain.npy", mmap_mode="r")
    np.testing.assert_array_equal(np.asarray(back), ram.chain)
    np.testing.assert_array_equal(disk.parameters, ram.parameters)
    stopped = d3d.Run(cube, inst, max_iterations=40, chain_file=str(tmp_path / "stop"),
                      variance=var, seed=3, keep_one_in=2, min_acceptance_rate=0.999,
                      jump_amplitude=5.0)
    assert stopped.iterations_done < 40
    n_valid = (stopped.iterations_done - 1) // 2 + 1
    assert np.isnan(stopped.chain[n_valid:]).all() and not np.isnan(stopped.chain[:n_valid]).any()
