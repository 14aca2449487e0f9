"""
k_mh_chain: whole sweeps of a small part in ONE launch of persistent workgroups, one per
lattice slot, colour classes handed from workgroup to workgroup through per-slot epoch flags
and write-through (sc1) residual / G-row traffic (csrc/d3d_kernels.h).  It replaces the
121 dependent colour launches of lib/run.py:367-519's sweep wherever a part's slots are all
resident at once (configs 1 and 2, the parts of a tiled 300x300 chain).

The kernel keeps a slot's window in registers and groups the window sums by window column:
another order of the same sums than the colour launches', so it agrees with them -- and,
update by update, with the oracle -- to rounding (1e-9, the tolerance of every device /
oracle comparison), with equal accepted counts.  It is bit-identical to ITSELF however the
sweeps are cut into launches (the pending layer and the parameters carry over), which is
what a stale line in any hand-off would break: partitioned and unpartitioned contexts,
masks, odd depths, rectangular FSFs, uniform variance.
"""
import numpy as np
import pytest

from deconv3d_amd import _lib, tiling
from oracle import deconv3d_oracle as O
from tests.cases import make_case
from tests.tiling_oracle import sweep_in_part_order

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not _lib.has_experiments() if _lib.device_count() else True,
                                 reason="k_mh_chain is only in a `make EXPERIMENTS=1` build "
                                        "(tools/build_experiments.sh; DECONV3D_HIP_LIB=...)")]


def run(case, chain, sweeps=3, per_call=None, lay=None, uniform=False, keep=None, extra=None):
    D, H, W = case["D"], case["H"], case["W"]
    opts = {"mh_chain": chain}
    opts.update(extra or {})
    with _lib.Engine((D, H, W), case["fsf"].shape, options=opts) as eng:
        eng.set_taps(case["fsf"], case["lsf"])
        if uniform:
            eng.set_data(case["data"], None, var_scalar=float(np.median(case["var"])), mask=case["mask"])
        else:
            eng.set_data(case["data"], case["var"], mask=case["mask"])
        if lay is not None:
            tiling.apply_parts(eng, lay)
        used = eng.get_option("chain_parts")
        eng.set_params(case["init"])
        eng.mh_config(case["min_b"], case["max_b"], 0.1, 35.0, seed=77, refresh_every=0)
        err0 = eng.residual()
        acc = 0
        chain_out = None
        if keep:
            chain_out = np.full((sweeps // keep + 1, H, W, 3), np.nan)
        s = 1
        while s <= sweeps:
            n = min(per_call or sweeps, sweeps - s + 1)
            acc += eng.mh_sweeps(n, s, keep or 1, chain=chain_out)
            s += n
        return dict(params=eng.get_params(), dlog=eng.get_dlog(),
                    err=eng.download_slot(_lib.SLOT_ERR), acc=np.int64(acc), used=used,
                    err0=err0, chain=chain_out)


def same(a, b):
    for k in ("params", "dlog", "err", "acc"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)


def close(a, b, mask):
    """chain kernel vs colour launches: the same sums in another grouping"""
    live = np.asarray(mask) == 1
    np.testing.assert_allclose(a["params"][live], b["params"][live], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(a["dlog"][live], b["dlog"][live], rtol=1e-9,
                               atol=1e-9 * max(np.abs(b["dlog"][live]).max(), 1e-300))
    assert np.max(np.abs(a["err"] - b["err"])) <= 1e-11 * np.max(np.abs(b["err"]))
    assert a["acc"] == b["acc"]


@pytest.mark.parametrize("name", ["c1", "odd_depth", "d30", "asym", "moffat", "nolsf", "rect_fsf",
                                  "tiny", "tile_a", "tile_b"])
def test_chain_kernel_matches_the_colour_launches_and_itself(name):
    case = make_case(name)
    ref = run(case, 0)
    one = run(case, 1)                 # three sweeps in ONE launch
    assert ref["used"] == 0 and one["used"] == 1
    close(one, ref, case["mask"])
    same(run(case, 1, per_call=1), one)   # one launch per sweep: the pending layer carries over
    same(run(case, 1), one)               # and it is deterministic


@pytest.mark.parametrize("name", ["c1", "moffat"])
def test_chain_kernel_with_uniform_variance_and_saved_sweeps(name):
    case = make_case(name)
    u1 = run(case, 1, uniform=True)
    assert u1["used"] == 1
    close(u1, run(case, 0, uniform=True), case["mask"])
    a, b = run(case, 1, sweeps=7, keep=2), run(case, 0, sweeps=7, keep=2)
    close(a, b, case["mask"])
    live = case["mask"] == 1
    np.testing.assert_allclose(a["chain"][1:][:, live], b["chain"][1:][:, live], rtol=1e-9, atol=1e-9)
    same(run(case, 1, sweeps=7, keep=2, per_call=3), a)


def test_config2_shape_against_the_oracle_and_the_colour_launches():
    """BASELINE config 2's shape (64x64x64, Moffat 11x11, 17-tap LSF, heteroscedastic
    variance): two sweeps in one launch, update by update against the oracle."""
    import bench as B
    D, H, W, fs = B.WORKLOADS["c2_64x64x64"]
    fsf, lsf = B.build_taps(D, fs)
    with _lib.Engine((D, H, W), fsf.shape) as eng:
        eng.set_taps(fsf, lsf)
        data, var, truth, init, min_b, max_b = B.synthetic_inputs(eng, D, H, W, fsf, 12345)
    mask = np.ones((H, W))
    mask[5, 60] = mask[31, 32] = 0
    case = dict(D=D, H=H, W=W, fsf=fsf, lsf=lsf, data=data, var=var, mask=mask, init=init,
                min_b=min_b, max_b=max_b)
    got = run(case, 1, sweeps=2)
    assert got["used"] == 1
    close(got, run(case, 0, sweeps=2), mask)
    st = O.MHState(data, var, mask, fsf, lsf, init, min_b, max_b, 0.1, 35.0, 77, err=got["err0"])
    for s in (1, 2):
        O.mh_sweep(st, s)
    live = mask == 1
    np.testing.assert_allclose(got["params"][live], st.params[live], rtol=1e-9, atol=1e-9)
    assert got["acc"] == st.accepted
    assert np.max(np.abs(got["err"] - st.err)) <= 1e-11 * np.max(np.abs(st.err))


@pytest.mark.parametrize("grid", [(2, 1), (2, 2)])
def test_partitioned_128_channel_context_takes_the_wide_chain_form(grid):
    """The parts of a tiled 128-channel chain: the chain kernel agrees with the colour
    launches (their wide form: eleven streaming wavefronts per window) and follows the
    oracle in part order."""
    import bench as B
    D, H, W, fs = 128, 60, 70, 11
    fsf, lsf = B.build_taps(D, fs)
    with _lib.Engine((D, H, W), fsf.shape) as eng:
        eng.set_taps(fsf, lsf)
        data, var, truth, init, min_b, max_b = B.synthetic_inputs(eng, D, H, W, fsf, 3)
    mask = np.ones((H, W))
    mask[7, 11] = mask[30, 35] = mask[59, 0] = 0
    case = dict(D=D, H=H, W=W, fsf=fsf, lsf=lsf, data=data, var=var, mask=mask, init=init,
                min_b=min_b, max_b=max_b)
    lay = tiling.TileLayout(H, W, fs, fs, *grid)
    got = run(case, 1, sweeps=2, lay=lay)
    ref = run(case, 0, sweeps=2, lay=lay)
    assert got["used"] == len(lay.all_parts()) and ref["used"] == 0
    close(got, ref, mask)
    st = O.MHState(data, var, mask, fsf, lsf, init, min_b, max_b, 0.1, 35.0, 77, err=got["err0"])
    for s in (1, 2):
        sweep_in_part_order(st, lay, s)
    live = mask == 1
    np.testing.assert_allclose(got["params"][live], st.params[live], rtol=1e-9, atol=1e-9)
    assert got["acc"] == st.accepted
    assert np.max(np.abs(got["err"] - st.err)) <= 1e-11 * np.max(np.abs(st.err))


def test_two_contexts_with_different_options_coexist():
    """options belong to a context, not to the process."""
    case = make_case("moffat")
    D, H, W = case["D"], case["H"], case["W"]
    engs = []
    for opts in ({"mh_chain": 0, "mh_layers": 2}, {"mh_chain": 0, "mh_layers": 3}, {"mh_chain": 1}):
        e = _lib.Engine((D, H, W), case["fsf"].shape, options=opts)
        e.set_taps(case["fsf"], case["lsf"])
        e.set_data(case["data"], case["var"], mask=case["mask"])
        e.set_params(case["init"])
        e.mh_config(case["min_b"], case["max_b"], 0.1, 35.0, seed=5, refresh_every=0)
        engs.append(e)
    try:
        assert [e.mh_layers() for e in engs] == [2, 3, 1]
        assert [e.get_option("chain_parts") for e in engs] == [0, 0, 1]
        for s in (1, 2, 3):                  # interleaved: every context keeps its own setting
            for e in engs:
                e.mh_sweeps(1, s)
        outs = [(e.get_params(), e.download_slot(_lib.SLOT_ERR)) for e in engs]
        np.testing.assert_array_equal(outs[1][0], outs[0][0])      # layer depths: bit for bit
        np.testing.assert_array_equal(outs[1][1], outs[0][1])
        live = case["mask"] == 1
        np.testing.assert_allclose(outs[2][0][live], outs[0][0][live], rtol=1e-9, atol=1e-9)
        with pytest.raises(ValueError):
            engs[0].set_option("no_such_option", 1)
        with pytest.raises(ValueError):
            engs[0].set_option("mh_layers", 7)
    finally:
        for e in engs:
            e.close()


def test_tiles_running_the_chain_kernel_equal_the_partitioned_single_context():
    """Tiling keeps its guarantee under the chain kernel: tile contexts (loop-back on one
    GPU) and the single context given the same parts, both with mh_chain = 1, agree to the
    last bit -- slot grids are anchored to GLOBAL colour classes and lattice points."""
    import bench as B
    D, H, W, fs = 128, 60, 70, 11
    fsf, lsf = B.build_taps(D, fs)
    with _lib.Engine((D, H, W), fsf.shape) as eng:
        eng.set_taps(fsf, lsf)
        data, var, truth, init, min_b, max_b = B.synthetic_inputs(eng, D, H, W, fsf, 3)
    mask = np.ones((H, W))
    mask[7, 11] = mask[30, 35] = 0
    lay = tiling.TileLayout(H, W, fs, fs, 2, 2)
    with _lib.Engine((D, H, W), fsf.shape, options={"mh_chain": 1}) as ref:
        ref.set_taps(fsf, lsf)
        ref.set_data(data, var, mask=mask)
        tiling.apply_parts(ref, lay)
        assert ref.get_option("chain_parts") == len(lay.all_parts())
        ref.set_params(init)
        ref.mh_config(min_b, max_b, 0.1, 35.0, seed=9, refresh_every=0)
        err0 = ref.residual()
        acc = ref.mh_sweeps(2, 1)
        ref_params, ref_err = ref.get_params(), ref.download_slot(_lib.SLOT_ERR)
    engines = [tiling.make_tile_engine(lay, r, data, var, mask, fsf, lsf, init, min_b, max_b, 0.1, 35.0,
                                       9, err=err0, options={"mh_chain": 1}) for r in range(lay.n)]
    try:
        assert all(e.get_option("chain_parts") >= 1 for e in engines)
        tables = [tiling.plan_tables(lay, r) for r in range(lay.n)]
        for s in (1, 2):
            tiling.sweep_loopback(engines, lay, tables, s, device_copy=True)
        live = mask == 1
        for r in range(lay.n):
            (y0, y1, x0, x1), p = tiling.gather_params(lay, r, engines[r])
            m = live[y0:y1, x0:x1]
            np.testing.assert_array_equal(p[m], ref_params[y0:y1, x0:x1][m])
            uy0, uy1, ux0, ux1 = lay.used(r)
            ry0, _, rx0, _ = lay.region(r)
            err = engines[r].download_slot(_lib.SLOT_ERR)[:, uy0 - ry0:uy1 - ry0, ux0 - rx0:ux1 - rx0]
            np.testing.assert_array_equal(err, ref_err[:, uy0:uy1, ux0:ux1])
        assert sum(e.mh_accepted() for e in engines) == acc
    finally:
        for e in engines:
            e.close()
