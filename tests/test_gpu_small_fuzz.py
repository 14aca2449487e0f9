"""
Randomised problems through k_mh_small (csrc/d3d_mh_small.h, round 4): the colour launches that
do not fill the chip -- relative position tables, the sweep's line table, the decision on the
channel wavefronts; the wide form for partitioned 65..128-channel contexts -- against the
round-3 kernels the same launches took before (option mh_small = 0): the same chain BIT FOR
BIT (parameters, carried residual, log-ratio map, accepted count).  Drawn per seed: depth,
footprint, FSF size (square or not) and kind, LSF kind, mask, variance kind, zig-zag, a
partition into one to three row or column parts with their phases, refresh cadence, sweeps.
`D3D_TEST_RANDOM_SMALL=N` draws N seeds (default 24).
"""
import os

import numpy as np
import pytest

from deconv3d_amd import _lib
from oracle import deconv3d_oracle as O

pytestmark = pytest.mark.gpu


def draw(seed):
    rng = np.random.default_rng(5000 + seed)
    D = int(rng.choice([8, 16, 21, 30, 32, 48, 64, 65, 100, 127, 128, 200, 256]))
    fh = int(rng.choice([3, 5, 7, 9, 11, 13, 15]))
    fw = fh if rng.random() < 0.7 else int(rng.choice([3, 5, 7, 9, 11]))
    H, W = int(rng.integers(1, 70)), int(rng.integers(1, 70))
    kind = rng.choice(["moffat", "random", "gauss"])
    if kind == "moffat" and fh == fw:
        fsf = O.moffat_cropped(fh, float(rng.uniform(1.5, 3.5)), 2.5)
    else:
        fsf = rng.random((fh, fw)) if kind != "gauss" else np.outer(np.hanning(fh + 2)[1:-1], np.hanning(fw + 2)[1:-1])
        fsf = fsf / fsf.sum()
    lkind = rng.choice(["none", "muse", "gauss", "wide"])
    lsf = {"none": lambda: None, "muse": lambda: O.muse_like_lsf(D),
           "gauss": lambda: O.gaussian_lsf_vector(D, float(rng.uniform(0.3, 1.0))),
           "wide": lambda: O.gaussian_lsf_vector(D, float(rng.uniform(2.0, 3.0)))}[lkind]()
    truth = np.dstack((1.0 + 9.0 * rng.random((H, W)), D * (0.2 + 0.6 * rng.random((H, W))),
                       0.6 + 2.0 * rng.random((H, W))))
    mask = (rng.random((H, W)) < rng.choice([1.0, 0.9, 0.6])).astype(float)
    if mask.sum() == 0:
        mask[rng.integers(0, H), rng.integers(0, W)] = 1
    sigma = 0.3
    data = rng.normal(0.0, sigma, size=(D, H, W)) + truth[..., 0][None] * 0.05
    uniform = rng.random() < 0.3
    var = np.full((D, H, W), sigma ** 2) if uniform else sigma ** 2 * (0.5 + rng.random((D, H, W)))
    init = truth * (0.8 + 0.4 * rng.random((H, W, 3)))
    # partition: none, or 2-3 strips along y or x with phases in order
    parts = None
    if rng.random() < 0.5 and max(H, W) >= 6:
        along_y = H >= W
        n = H if along_y else W
        k = int(rng.integers(2, 4))
        cuts = sorted(set(int(v) for v in rng.integers(1, n, size=k - 1)))
        edges = [0] + cuts + [n]
        rects = [(a, b, 0, W) if along_y else (0, H, a, b) for a, b in zip(edges[:-1], edges[1:])]
        phases = sorted(int(v) for v in rng.integers(0, 3, size=len(rects)))
        parts = (rects, phases)
    opts = {"mh_zigzag": int(rng.integers(0, 2))}
    return dict(D=D, H=H, W=W, fsf=fsf, lsf=lsf, data=data, var=var, uniform=uniform, mask=mask, init=init,
                parts=parts, opts=opts, refresh=int(rng.choice([0, 2, 3])), seed=int(rng.integers(1, 1000)),
                sweeps=int(rng.integers(2, 5)))


def run_chain(c, small):
    shape = (c["D"], c["H"], c["W"])
    with _lib.Engine(shape, c["fsf"].shape, options=dict(c["opts"], mh_small=small)) as eng:
        eng.set_taps(c["fsf"], c["lsf"])
        if c["uniform"]:
            eng.set_data(c["data"], None, var_scalar=float(c["var"].flat[0]), mask=c["mask"])
        else:
            eng.set_data(c["data"], c["var"], mask=c["mask"])
        if c["parts"]:
            eng.set_parts(*c["parts"])
        eng.set_params(c["init"])
        mn = np.array([0.0, 0.0, 0.3])
        mx = np.array([30.0, c["D"] - 1.0, 6.0])
        eng.mh_config(mn, mx, 0.1, 900.0, seed=c["seed"], refresh_every=c["refresh"])
        acc = eng.mh_sweeps(c["sweeps"], 1)
        acc += eng.mh_sweeps(1, c["sweeps"] + 1)
        small_parts = eng.get_option("small_parts")
        return (eng.get_params(), eng.download_slot(_lib.SLOT_ERR), eng.get_dlog(), np.array([acc])), small_parts


@pytest.mark.parametrize("seed", range(int(os.environ.get("D3D_TEST_RANDOM_SMALL", "24"))))
def test_random_small_launch_problem_is_bit_identical_to_round3s_kernels(seed):
    c = draw(seed)
    new, on_small = run_chain(c, 1)
    old, on_old = run_chain(c, 0)
    assert on_old == 0
    if on_small == 0:                          # (a 3x3 FSF on a 69x69 footprint: 529 windows per launch)
        pytest.skip("every launch of this draw fills the chip: not k_mh_small's case")
    for a, b in zip(new, old):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("seed", range(int(os.environ.get("D3D_TEST_RANDOM_BATCH", "12"))))
def test_random_batched_chains_are_the_chains_they_would_be_alone(seed):
    """d3d_mh_sweeps_batch on the same random problems (unpartitioned): R chains of one geometry --
    own data, start and seed each -- advanced by ONE launch per colour class are, bit for bit,
    the chains their contexts produce alone (k_mh_small<..., BATCH> while the joint launch does
    not fill the chip, k_mh_ws's batched form once it does), over two calls."""
    from deconv3d_amd import ensemble
    c = draw(seed)
    rng = np.random.default_rng(9000 + seed)
    R = int(rng.integers(2, 9))
    shape = (c["D"], c["H"], c["W"])
    mn, mx = np.array([0.0, 0.0, 0.3]), np.array([30.0, c["D"] - 1.0, 6.0])

    def make(r):
        eng = _lib.Engine(shape, c["fsf"].shape, options=c["opts"])
        eng.set_taps(c["fsf"], c["lsf"])
        data = c["data"] * (1.0 + 0.1 * r)
        if c["uniform"]:
            eng.set_data(data, None, var_scalar=float(c["var"].flat[0]) * (1.0 + 0.2 * r), mask=c["mask"])
        else:
            eng.set_data(data, c["var"] * (1.0 + 0.05 * r), mask=c["mask"])
        init = c["init"].copy()
        init[..., 2] = np.clip(init[..., 2] + 0.05 * r, 0.3, 6.0)
        eng.set_params(init)
        eng.mh_config(mn, mx, 0.1, 900.0, seed=c["seed"] + r, refresh_every=c["refresh"])
        return eng

    alone = []
    for r in range(R):
        with make(r) as eng:
            acc = eng.mh_sweeps(c["sweeps"], 1) + eng.mh_sweeps(2, c["sweeps"] + 1)
            alone.append((eng.get_params(), eng.download_slot(_lib.SLOT_ERR), eng.get_dlog(), acc))
    engs = [make(r) for r in range(R)]
    try:
        a1 = ensemble.sweep_chains_batched(engs, c["sweeps"], 1)
        a2 = ensemble.sweep_chains_batched(engs, 2, c["sweeps"] + 1)
        for r, eng in enumerate(engs):
            np.testing.assert_array_equal(eng.get_params(), alone[r][0])
            np.testing.assert_array_equal(eng.download_slot(_lib.SLOT_ERR), alone[r][1])
            np.testing.assert_array_equal(eng.get_dlog(), alone[r][2])
            assert a1[r] + a2[r] == alone[r][3]
    finally:
        for e in engs:
            e.close()
