"""
Whole sweeps at the BASELINE shapes, update by update against the oracle
(lib/run.py:367-519 restated in oracle.mh_update): the kernels that actually run at
those shapes are compared with the CPU restatement itself, not only with each other.

  * config 3, 300x300x128, default kernel selection (k_mh_ws with two pending layers,
    zig-zag walk, virtual border positions);
  * the same cube partitioned for 8x1 ranks, in part order: the wide form of the small
    colour launches (k_mh_ws<704>: eleven streaming wavefronts per window), and that form
    against the 256-thread one;
  * a chip-filling cube with the beyond-the-Infinity-Cache policy forced on
    (non-temporal 1/variance loads, write-through residual stores) against the
    default policy, bit for bit, and against the oracle;
  * the zig-zag walk switched off;
  * a 264-channel cube whose launches fill the chip: the 512-thread form of k_mh_ws, both
    cache policies.

The oracle is fed the DEVICE's initial residual (the forward model has its own tests),
so a 90 000-update sweep costs about 20 s of numpy on the GPU box's host.
Tolerances: parameters and the log-ratio map rel 1e-9 (ulp-level differences of
tan / exp / erfcinv between libm and ocml), carried residual 1e-11 of its peak,
accepted counts equal.
"""
import numpy as np
import pytest

from deconv3d_amd import _lib, tiling
from oracle import deconv3d_oracle as O
from tests.tiling_oracle import sweep_in_part_order

pytestmark = pytest.mark.gpu

SEED = 12345


def build(D, H, W, fs, options=None):
    import bench as B
    fsf, lsf = B.build_taps(D, fs)
    eng = _lib.Engine((D, H, W), fsf.shape, options=options)
    eng.set_taps(fsf, lsf)
    data, var, truth, init, min_b, max_b = B.synthetic_inputs(eng, D, H, W, fsf, SEED)
    mask = np.ones((H, W))
    mask[17, min(200, W - 1)] = mask[H // 2 + 1, W // 2 - 1] = mask[H // 2, W // 2] = 0
    eng.set_data(data, var, mask=mask)
    return eng, dict(D=D, H=H, W=W, fsf=fsf, lsf=lsf, data=data, var=var, mask=mask, init=init,
                     min_b=min_b, max_b=max_b, ra=float(max_b[0] ** 2))


def start(eng, pb, seed=SEED):
    eng.set_params(pb["init"])
    eng.mh_config(pb["min_b"], pb["max_b"], 0.1, pb["ra"], seed=seed, refresh_every=0)
    return eng.residual()


def oracle_state(pb, err0, seed=SEED):
    return O.MHState(pb["data"], pb["var"], pb["mask"], pb["fsf"], pb["lsf"], pb["init"],
                     pb["min_b"], pb["max_b"], 0.1, pb["ra"], seed, err=err0)


def assert_matches_oracle(eng, st, accepted, pb):
    live = pb["mask"] == 1
    params = eng.get_params()
    np.testing.assert_allclose(params[live], st.params[live], rtol=1e-9, atol=1e-9)
    np.testing.assert_array_equal(params[~live], pb["init"][~live])     # lib/run.py:553-566
    dlog = eng.get_dlog()
    np.testing.assert_allclose(dlog[live], st.dlog[live], rtol=1e-9,
                               atol=1e-9 * np.abs(st.dlog[live]).max())
    assert accepted == st.accepted
    err = eng.download_slot(_lib.SLOT_ERR)      # flushes the pending layers
    assert np.max(np.abs(err - st.err)) <= 1e-11 * np.max(np.abs(st.err))


def test_config3_full_sweep_update_by_update_against_the_oracle():
    """BASELINE config 3 as bench.py runs it: 300x300x128, Moffat 11x11, 17-tap LSF,
    heteroscedastic variance, default kernel selection; one sweep = 89 997 updates."""
    eng, pb = build(128, 300, 300, 11)
    with eng:
        assert eng.mh_layers() == 2 and not eng.variance_is_uniform()
        err0 = start(eng, pb)
        st = oracle_state(pb, err0)
        accepted = eng.mh_sweeps(1, 1)
        assert O.mh_sweep(st, 1) == int(pb["mask"].sum())
        assert_matches_oracle(eng, st, accepted, pb)


def test_config4_partitioned_sweep_in_part_order_against_the_oracle():
    """The 300x300x128 cube with the parts of an 8x1 tiling on one context (the scan
    order a tiled chain has, tiling.apply_parts): its small colour launches run the
    wide form (eleven streaming wavefronts per window).  One sweep against the oracle in (phase, part, colour) order; then
    the 256-thread form (option mh_wide = 0) must agree to rounding -- another grouping
    of the window sums, not another algorithm."""
    lay = tiling.TileLayout(300, 300, 11, 11, 8, 1)
    outs = []
    for wide in (1, 0):
        eng, pb = build(128, 300, 300, 11, options={"mh_wide": wide})
        with eng:
            tiling.apply_parts(eng, lay)
            err0 = start(eng, pb)
            accepted = eng.mh_sweeps(1, 1)
            if wide:
                st = oracle_state(pb, err0)
                sweep_in_part_order(st, lay, 1)
                assert_matches_oracle(eng, st, accepted, pb)
            outs.append((eng.get_params(), eng.get_dlog(), eng.download_slot(_lib.SLOT_ERR), accepted))
    live = pb["mask"] == 1
    np.testing.assert_allclose(outs[0][0][live], outs[1][0][live], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(outs[0][1][live], outs[1][1][live], rtol=1e-9,
                               atol=1e-9 * np.abs(outs[0][1][live]).max())
    assert np.max(np.abs(outs[0][2] - outs[1][2])) <= 1e-11 * np.max(np.abs(outs[0][2]))
    assert outs[0][3] == outs[1][3]
    assert not np.array_equal(outs[0][2], outs[1][2])      # the two forms did run


def test_beyond_cache_policy_is_bit_identical_and_matches_the_oracle():
    """Non-temporal 1/variance loads and write-through residual stores (the policy of a
    context whose working set exceeds the 256 MiB Infinity Cache, option mh_nt_ivar)
    forced on for a 300x300x16 cube whose colour launches fill the chip: same bytes,
    same results as the default policy -- and both equal the oracle."""
    outs = []
    for nt in (0, 1):
        eng, pb = build(16, 300, 300, 11, options={"mh_nt_ivar": nt})
        with eng:
            assert eng.mh_layers() == 2          # chip-filling launches: the NTV kernels' family
            assert eng.get_option("mh_nt_ivar") == nt
            err0 = start(eng, pb)
            accepted = eng.mh_sweeps(2, 1)
            if nt:
                st = oracle_state(pb, err0)
                for s in (1, 2):
                    O.mh_sweep(st, s)
                assert_matches_oracle(eng, st, accepted, pb)
            outs.append((eng.get_params(), eng.get_dlog(), eng.download_slot(_lib.SLOT_ERR),
                         np.int64(accepted)))
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)


def test_512_thread_kernel_with_chip_filling_launches_matches_the_oracle():
    """257 .. 512 channels: k_mh_ws with 512 streaming threads, two pending layers, four
    positions in flight -- the variants of launches that fill the chip (>= 512 windows per
    colour class), with the default cache policy and with the beyond-cache one forced on
    (non-temporal 1/variance, write-through residual): bit-identical to each other, and one
    whole sweep equal to the oracle."""
    outs = []
    for nt in (0, 1):
        eng, pb = build(264, 253, 253, 11, options={"mh_nt_ivar": nt})
        with eng:
            assert eng.mh_layers() == 2
            err0 = start(eng, pb)
            accepted = eng.mh_sweeps(1, 1)
            if nt:
                st = oracle_state(pb, err0)
                O.mh_sweep(st, 1)
                assert_matches_oracle(eng, st, accepted, pb)
            outs.append((eng.get_params(), eng.get_dlog(), eng.download_slot(_lib.SLOT_ERR),
                         np.int64(accepted)))
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)


def test_z_blocked_kernels_with_chip_filling_launches():
    """Beyond 512 channels with launches that fill the chip (1100 channels = 5 blocks, 121
    windows per colour class: 605 workgroups): the two-layer form of the z-blocked kernels under
    both cache policies -- bit-identical --, against the ORACLE update by update (one sweep),
    against the thread-looped kernel of deep cubes (mh_zblocks = 0) to rounding, and the carried
    residual against the one rebuilt from the parameters."""
    outs = []
    for opts in ({"mh_nt_ivar": 0}, {"mh_nt_ivar": 1}, {"mh_zblocks": 0}):
        eng, pb = build(1100, 112, 112, 11, options=opts)
        with eng:
            if "mh_nt_ivar" in opts:
                assert eng.mh_layers() == 2
            err0 = start(eng, pb)
            if opts == {"mh_nt_ivar": 0}:
                # round 4: this variant against the ORACLE itself, update by
                # update -- one sweep of 12 541 updates, ~1 MB of window each
                st = oracle_state(pb, err0)
                accepted = eng.mh_sweeps(1, 1)
                O.mh_sweep(st, 1)
                assert_matches_oracle(eng, st, accepted, pb)
                start(eng, pb)
            accepted = eng.mh_sweeps(3, 1)
            carried = eng.download_slot(_lib.SLOT_ERR)
            params = eng.get_params()
            fresh = eng.residual()
            assert np.max(np.abs(carried - fresh)) <= 1e-11 * np.max(np.abs(fresh))
            outs.append((params, eng.get_dlog(), carried, np.int64(accepted)))
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a, b)
    live = pb["mask"] == 1
    np.testing.assert_allclose(outs[0][0][live], outs[2][0][live], rtol=1e-9, atol=1e-9)
    assert np.max(np.abs(outs[0][2] - outs[2][2])) <= 1e-11 * np.max(np.abs(outs[2][2]))
    assert outs[0][3] == outs[2][3]


@pytest.mark.parametrize("shape", [(64, 64, 64), (128, 150, 90)])
def test_zigzag_walk_off_and_on_match_the_oracle(shape):
    """Option mh_zigzag: every other colour class walks its window positions backwards.
    Both orders are the same sums in another order: each matches the oracle, and they
    agree with each other to rounding, not bit for bit."""
    D, H, W = shape
    outs = []
    for zz in (1, 0):
        eng, pb = build(D, H, W, 11, options={"mh_zigzag": zz})
        with eng:
            err0 = start(eng, pb, seed=99)
            st = oracle_state(pb, err0, seed=99)
            accepted = eng.mh_sweeps(2, 1)
            for s in (1, 2):
                O.mh_sweep(st, s)
            assert_matches_oracle(eng, st, accepted, pb)
            outs.append(eng.download_slot(_lib.SLOT_ERR))
    assert np.max(np.abs(outs[0] - outs[1])) <= 1e-11 * np.max(np.abs(outs[0]))
    assert not np.array_equal(outs[0], outs[1])
