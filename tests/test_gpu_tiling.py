"""
GPU tests of the spatial tiling (BASELINE config 4) on ONE device: every tile is
its own device context on the same GPU (loop-back), halos copied device to device
or staged through the host.  The tiled chain -- global colour classes, Philox
keyed by global spaxel index, parts / phases, bulk halo copies of residual cells --
is BIT-IDENTICAL to a single context given the same parts, at test sizes and at
the full 300x300x128 of config 4 (2x2, 2x4 and the default row strips).  The RCCL
transport is exercised as far as one GPU allows: a one-rank communicator sending
to itself.
"""
import numpy as np
import pytest

from deconv3d_amd import _lib, tiling
from oracle import deconv3d_oracle as O
from tests.cases import make_case
from tests.tiling_oracle import sweep_in_part_order

pytestmark = pytest.mark.gpu


def single_context(case, lay, ra, seed, sweeps, refresh_every=0):
    D, H, W = case["D"], case["H"], case["W"]
    with _lib.Engine((D, H, W), case["fsf"].shape) as ref:
        ref.set_taps(case["fsf"], case["lsf"])
        ref.set_data(case["data"], case["var"], mask=case["mask"])
        if lay is not None:
            tiling.apply_parts(ref, lay)
        ref.set_params(case["init"])
        ref.mh_config(case["min_b"], case["max_b"], 0.1, ra, seed=seed, refresh_every=refresh_every)
        err0 = ref.residual()
        accepted = ref.mh_sweeps(sweeps, 1)
        return ref.get_params(), ref.download_slot(_lib.SLOT_ERR), accepted, err0


def tiled_contexts(case, lay, ra, seed, err0, refresh_every=0):
    return [tiling.make_tile_engine(lay, r, case["data"], case["var"], case["mask"], case["fsf"],
                                    case["lsf"], case["init"], case["min_b"], case["max_b"], 0.1,
                                    ra, seed, err=err0, refresh_every=refresh_every)
            for r in range(lay.n)]


def compare(case, lay, engines, ref_params, ref_err, exact=True):
    live = case["mask"] == 1
    for r in range(lay.n):
        (y0, y1, x0, x1), p = tiling.gather_params(lay, r, engines[r])
        m = live[y0:y1, x0:x1]
        uy0, uy1, ux0, ux1 = lay.used(r)
        ry0, _, rx0, _ = lay.region(r)
        err = engines[r].download_slot(_lib.SLOT_ERR)[:, uy0 - ry0:uy1 - ry0, ux0 - rx0:ux1 - rx0]
        if exact:
            np.testing.assert_array_equal(p[m], ref_params[y0:y1, x0:x1][m])
            np.testing.assert_array_equal(err, ref_err[:, uy0:uy1, ux0:ux1])
        else:
            np.testing.assert_allclose(p[m], ref_params[y0:y1, x0:x1][m], rtol=1e-8, atol=1e-8)
            np.testing.assert_allclose(err, ref_err[:, uy0:uy1, ux0:ux1], rtol=0,
                                       atol=1e-9 * np.abs(ref_err).max())


@pytest.mark.parametrize("name,grid,device_copy", [
    ("tile_a", (2, 2), True), ("tile_a", (4, 1), True), ("tile_a", (1, 2), False),
    ("tile_b", (2, 3), True), ("tile_b", (2, 1), False), ("c1", (1, 1), True),
    ("tile_deep", (2, 1), True), ("tile_deep", (2, 2), True)])
def test_tiled_chain_is_bit_identical_to_single_context(name, grid, device_copy):
    case = make_case(name)
    fh, fw = case["fsf"].shape
    ra, seed, sweeps = 35.0, 77, 3
    lay = tiling.TileLayout(case["H"], case["W"], fh, fw, *grid)
    ref_params, ref_err, accepted, err0 = single_context(case, lay, ra, seed, sweeps)
    assert accepted > 0
    engines = tiled_contexts(case, lay, ra, seed, err0)
    try:
        tables = [tiling.plan_tables(lay, r) for r in range(lay.n)]
        for s in range(1, sweeps + 1):
            tiling.sweep_loopback(engines, lay, tables, s, device_copy=device_copy)
        compare(case, lay, engines, ref_params, ref_err)
        assert sum(e.mh_accepted() for e in engines) == accepted
    finally:
        for e in engines:
            e.close()


def test_single_context_with_parts_follows_the_oracle_in_part_order():
    """d3d_set_parts changes the scan order only: the partitioned single context
    equals the oracle run in (phase, part, colour) order."""
    case = make_case("tile_a")
    fh, fw = case["fsf"].shape
    lay = tiling.TileLayout(case["H"], case["W"], fh, fw, 2, 2)
    st = O.MHState(case["data"], case["var"], case["mask"], case["fsf"], case["lsf"],
                   case["init"], case["min_b"], case["max_b"], 0.1, 35.0, 77)
    for s in (1, 2):
        sweep_in_part_order(st, lay, s)
    params, err, accepted, _ = single_context(case, lay, 35.0, 77, 2)
    np.testing.assert_allclose(params, st.params, rtol=1e-9, atol=1e-9)
    assert np.max(np.abs(err - st.err)) <= 1e-11 * np.max(np.abs(st.err))
    assert accepted == st.accepted
    # and differs from the unpartitioned order (it IS another scan order)
    plain, _, _, _ = single_context(case, None, 35.0, 77, 2)
    assert not np.array_equal(plain, params)


def test_tiled_refresh_gathers_parameters_and_rebuilds_the_residual():
    """lib/run.py:521-534 in a tiled run (refresh every 2nd sweep): parameter gather,
    then each tile's own from-scratch residual -- rounding-level agreement with the
    single context (the convolution kernels sum in a position-dependent order)."""
    case = make_case("tile_b")
    fh, fw = case["fsf"].shape
    lay = tiling.TileLayout(case["H"], case["W"], fh, fw, 2, 2)
    ref_params, ref_err, _, _ = single_context(case, lay, 35.0, 5, 4, refresh_every=2)
    engines = tiled_contexts(case, lay, 35.0, 5, None, refresh_every=2)
    try:
        tables = [tiling.plan_tables(lay, r) for r in range(lay.n)]
        for s in range(1, 5):
            tiling.sweep_loopback(engines, lay, tables, s, device_copy=True, refresh=(s % 2 == 0))
        compare(case, lay, engines, ref_params, ref_err, exact=False)
    finally:
        for e in engines:
            e.close()


@pytest.mark.parametrize("grid", [(2, 2), (2, 4), (4, 1), (8, 1)])
def test_config4_full_size_tiled_chain_is_bit_identical(grid):
    """BASELINE config 4 at its real size: 300x300x128, Moffat 11x11, 17-tap LSF,
    heteroscedastic variance; four / eight tile contexts in loop-back on one GPU,
    two sweeps (180 000 updates), parameters and residual bit-identical."""
    import bench as B
    D, H, W, fs = B.WORKLOADS["c3_300x300x128"]
    fsf, lsf = B.build_taps(D, fs)
    with _lib.Engine((D, H, W), fsf.shape) as full:
        full.set_taps(fsf, lsf)
        data, var, truth, init, min_b, max_b = B.synthetic_inputs(full, D, H, W, fsf, 12345)
    mask = np.ones((H, W))
    mask[17, 200] = mask[151, 149] = mask[150, 150] = 0
    case = dict(D=D, H=H, W=W, fsf=fsf, lsf=lsf, data=data, var=var, mask=mask, init=init,
                min_b=min_b, max_b=max_b)
    lay = tiling.TileLayout(H, W, fs, fs, *grid)
    ra = float(max_b[0] ** 2)
    ref_params, ref_err, accepted, err0 = single_context(case, lay, ra, 12345, 2)
    engines = tiled_contexts(case, lay, ra, 12345, err0)
    try:
        tables = [tiling.plan_tables(lay, r) for r in range(lay.n)]
        for s in (1, 2):
            tiling.sweep_loopback(engines, lay, tables, s, device_copy=True)
        compare(case, lay, engines, ref_params, ref_err)
        assert sum(e.mh_accepted() for e in engines) == accepted
    finally:
        for e in engines:
            e.close()


def test_rccl_halo_exchange_one_rank_sends_to_itself():
    """The in-library RCCL transport as far as one GPU allows: a one-rank communicator
    (ncclCommInitRank) whose plan sends a rectangle of residual cells and one of the
    parameter map to ITSELF (pack -> ncclSend/ncclRecv in a group -> unpack on the
    context's stream)."""
    case = make_case("tile_a")
    D, H, W = case["D"], case["H"], case["W"]
    with _lib.Engine((D, H, W), case["fsf"].shape) as eng:
        eng.set_taps(case["fsf"], case["lsf"])
        eng.set_data(case["data"], case["var"], mask=case["mask"])
        eng.set_params(case["init"])
        err = eng.residual()
        eng.comm_init(1, 0, _lib.comm_unique_id())
        eng.halo_plan(0, [[0, 0, 2, 9, 3, 11, 20, 27, 10, 18],
                          [0, 1, 0, 4, 0, 26, 30, 34, 0, 26]])
        eng.halo_exchange(0)
        got = eng.download_slot(_lib.SLOT_ERR)
        want = err.copy()
        want[:, 20:27, 10:18] = err[:, 2:9, 3:11]
        np.testing.assert_array_equal(got, want)
        p = eng.get_params()
        wantp = np.array(case["init"])
        wantp[30:34] = case["init"][0:4]
        np.testing.assert_array_equal(p, wantp)
        eng.comm_destroy()


def test_mh_sweeps_refuses_a_tile_with_halo_plans_but_no_communicator():
    case = make_case("tile_a")
    lay = tiling.TileLayout(case["H"], case["W"], *case["fsf"].shape, 2, 1)
    eng = tiling.make_tile_engine(lay, 0, case["data"], case["var"], case["mask"], case["fsf"],
                                  case["lsf"], case["init"], case["min_b"], case["max_b"], 0.1,
                                  10.0, 1)
    try:
        with pytest.raises(RuntimeError):
            eng.mh_sweeps(1, 1)
    finally:
        eng.close()


def test_colour_counts_respect_ownership():
    case = make_case("tile_a")
    fh, fw = case["fsf"].shape
    lay = tiling.TileLayout(case["H"], case["W"], fh, fw, 2, 2)
    total = np.zeros(fh * fw, int)
    engines = tiled_contexts(case, lay, 10.0, 1, None)
    try:
        for e in engines:
            total += np.array([e.colour_count(c) for c in range(fh * fw)])
    finally:
        for e in engines:
            e.close()
    want = np.zeros(fh * fw, int)
    for y in range(case["H"]):
        for x in range(case["W"]):
            if case["mask"][y, x] == 1:
                want[(y % fh) * fw + (x % fw)] += 1
    np.testing.assert_array_equal(total, want)


def test_update_records_replay_on_another_context():
    """The finer-grained alternative to halo copies (d3d_mh_colour /
    d3d_export_updates / d3d_apply_updates): the 8-double records of one colour
    class, replayed on a second context, reproduce the first one's residual."""
    case = make_case("c1")
    D, H, W = case["D"], case["H"], case["W"]
    fh, fw = case["fsf"].shape

    def ctx():
        e = _lib.Engine((D, H, W), (fh, fw))
        e.set_taps(case["fsf"], case["lsf"])
        e.set_data(case["data"], case["var"], mask=case["mask"])
        e.set_params(case["init"])
        e.mh_config(case["min_b"], case["max_b"], 0.1, 35.0, seed=3, refresh_every=0)
        e.residual(fetch=False)
        return e
    a, b = ctx(), ctx()
    try:
        for colour in (0, 40, 80):
            a.mh_colour(colour, 1)
            cy, cx = divmod(colour, fw)
            ys, xs = np.nonzero(case["mask"][cy::fh, cx::fw] == 1)
            idx = (cy + ys * fh) * W + (cx + xs * fw)
            b.apply_updates(a.export_updates(idx.astype(np.int32)))
        np.testing.assert_array_equal(b.download_slot(_lib.SLOT_ERR), a.download_slot(_lib.SLOT_ERR))
        np.testing.assert_array_equal(b.get_params(), a.get_params())
    finally:
        a.close()
        b.close()


def test_mh_sweeps_runs_phases_and_rccl_exchanges_from_one_call():
    """d3d_mh_sweeps on a context with a communicator: phases, RCCL halo exchanges between
    them and the parameter gather + from-scratch residual, all queued from ONE host call.
    One GPU allows one rank, so the plans send every rectangle to the rank itself (onto
    the same cells): the chain must equal the partitioned context without a communicator."""
    case = make_case("tile_b")
    fh, fw = case["fsf"].shape
    lay = tiling.TileLayout(case["H"], case["W"], fh, fw, 2, 1)
    ref_params, ref_err, accepted, _ = single_context(case, lay, 35.0, 9, 4, refresh_every=2)
    D, H, W = case["D"], case["H"], case["W"]
    with _lib.Engine((D, H, W), (fh, fw)) as eng:
        eng.set_taps(case["fsf"], case["lsf"])
        eng.set_data(case["data"], case["var"], mask=case["mask"])
        tiling.apply_parts(eng, lay)
        for ph in lay.phases:
            t = lay.touched(0, ph) or lay.touched(1, ph)
            eng.halo_plan(ph, [[0, 0, *t, *t]])
        eng.halo_plan(_lib.PLAN_PARAMS, [[0, 1, 3, 9, 0, W, 3, 9, 0, W]])
        eng.set_params(case["init"])
        eng.mh_config(case["min_b"], case["max_b"], 0.1, 35.0, seed=9, refresh_every=2)
        eng.residual(fetch=False)
        with pytest.raises(RuntimeError):        # plans but no communicator yet
            eng.mh_sweeps(1, 1)
        eng.comm_init(1, 0, _lib.comm_unique_id())
        got = eng.mh_sweeps(4, 1)
        np.testing.assert_array_equal(eng.get_params(), ref_params)
        np.testing.assert_array_equal(eng.download_slot(_lib.SLOT_ERR), ref_err)
        assert got == accepted
        eng.comm_destroy()


GLOO_WORKER = r"""
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
from deconv3d_amd import tiling
from tests.cases import make_case

dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
case = make_case("tile_a")
fh, fw = case["fsf"].shape
lay = tiling.TileLayout(case["H"], case["W"], fh, fw, *tiling.tile_grid_for(world))
err0 = np.load(os.path.join(%(out)r, "err0.npy"))
eng = tiling.make_tile_engine(lay, rank, case["data"], case["var"], case["mask"], case["fsf"],
                              case["lsf"], case["init"], case["min_b"], case["max_b"], 0.1, 35.0, 77,
                              device=0, err=err0)
tables = tiling.plan_tables(lay, rank)
for s in (1, 2, 3):
    tiling.sweep_distributed(eng, lay, tables, s, dist, torch)
(y0, y1, x0, x1), p = tiling.gather_params(lay, rank, eng)
np.save(os.path.join(%(out)r, "params_%%d.npy" %% rank), p)
np.save(os.path.join(%(out)r, "rect_%%d.npy" %% rank), np.array([y0, y1, x0, x1]))
eng.close()
dist.barrier()
dist.destroy_process_group()
"""


def test_two_gloo_ranks_on_one_gpu_are_bit_identical(tmp_path):
    """The multi-process form of the tiled chain with device engines: two ranks (both on
    this GPU), halos staged through the host over gloo, against the partitioned single
    context."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    case = make_case("tile_a")
    lay = tiling.TileLayout(case["H"], case["W"], *case["fsf"].shape, *tiling.tile_grid_for(2))
    ref_params, _, _, err0 = single_context(case, lay, 35.0, 77, 3)
    np.save(tmp_path / "err0.npy", err0)
    script = tmp_path / "worker.py"
    script.write_text(GLOO_WORKER % {"root": root, "out": str(tmp_path)})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)]
    res = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    live = case["mask"] == 1
    for r in range(2):
        y0, y1, x0, x1 = np.load(tmp_path / ("rect_%d.npy" % r))
        m = live[y0:y1, x0:x1]
        np.testing.assert_array_equal(np.load(tmp_path / ("params_%d.npy" % r))[m],
                                      ref_params[y0:y1, x0:x1][m])


def test_bench_two_ranks_report_the_tiled_leg():
    """`bench.py --gpus 2` (no launcher; gloo, both ranks on this one GPU): the contract line
    is the ensemble's, and the config-4 leg -- one chain tiled over the same ranks, run by
    child processes the ranks started before touching the GPU -- arrives as
    `config4_tiled`; rank 0's CPU baselines stay out of an N > 1 line."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2",
                        "--backend", "gloo", "--workload", "c2_64x64x64", "--steps", "2",
                        "--warmup", "1", "--conv-iters", "2"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["value"] > 0
    assert "cpu_baseline" not in rec
    leg = rec["config4_tiled"]
    assert "error" not in leg, leg
    assert leg["n_gpus"] == 2 and leg["scaling"] == "strong" and leg["value"] > 0
    assert "tiled 2x1" in leg["config"]["parallelism"]
    assert 0.0 < leg["acceptance"] < 1.0
    # the leg verifies itself: gathered parameters == one context given the same parts
    # (bit for bit: at 64 channels the tiles rebuild their residual with the one-pass
    # convolution kernel, whose sums do not depend on where a tile starts)
    assert leg["bit_identical"] is True, leg
    assert leg["tiles"] == [2, 1] and leg["phases_per_sweep"] == 2
    assert leg["rccl_ranks"] is None            # gloo rehearsal: no RCCL communicator
    assert leg["halo_ms_per_sweep"] > 0.0
    # the compute side, rank by rank: sum over the two phases of the slowest rank's own time --
    # a floor of the measured sweep (here both ranks share one GPU, so it is well below it)
    assert leg["slowest_rank"] in (0, 1)
    assert 0.0 < leg["projected_critical_path_ms"] <= leg["ms_per_step"]
    assert sorted(leg["critical_path_by_phase"]) == ["0", "2"]


def test_a_larger_halo_plan_after_a_streamed_chain_leaves_the_streaming_buffers_alone():
    """d3d_halo_plan used to free the chain-streaming buffers (device
    snapshots, pinned ring, copy stream) when it grew the send buffer; the next streamed
    sweep then copied through freed memory.  Stream a chain, grow the plan, stream again."""
    case = make_case("tile_a")
    D, H, W = case["D"], case["H"], case["W"]
    with _lib.Engine((D, H, W), case["fsf"].shape) as eng, \
            _lib.Engine((D, H, W), case["fsf"].shape) as ref:
        for e in (eng, ref):
            e.set_taps(case["fsf"], case["lsf"])
            e.set_data(case["data"], case["var"], mask=case["mask"])
            e.set_params(case["init"])
            e.mh_config(case["min_b"], case["max_b"], 0.1, 35.0, seed=3, refresh_every=0)
        chain = np.full((7, H, W, 3), np.nan)
        want = np.full((7, H, W, 3), np.nan)
        eng.halo_plan(_lib.PLAN_PARAMS, [[0, 0, 0, 2, 0, 3, 0, 2, 0, 3]])      # small plan
        eng.mh_sweeps(3, 1, 1, chain=chain)
        eng.halo_plan(_lib.PLAN_PARAMS, [[0, 0, 0, H, 0, W, 0, H, 0, W]])      # grows both buffers
        eng.mh_sweeps(3, 4, 1, chain=chain)
        ref.mh_sweeps(3, 1, 1, chain=want)
        ref.mh_sweeps(3, 4, 1, chain=want)
        np.testing.assert_array_equal(chain[1:], want[1:])


def test_a_small_part_before_a_large_one_keeps_its_own_kernel_family():
    """the pending-layer depth -- and with it the kernel family -- is chosen per
    PART.  A part list whose first part is small (one layer) and whose second fills the
    chip (two layers) used to fail with 'internal: 2 pending layers for a 1-layer
    kernel'.  Against the oracle in part order."""
    from tests.tiling_oracle import part_order
    import bench as B
    D, H, W, fs = 8, 300, 300, 11
    fsf, lsf = B.build_taps(D, fs)
    with _lib.Engine((D, H, W), fsf.shape) as eng:
        eng.set_taps(fsf, lsf)
        data, var, truth, init, min_b, max_b = B.synthetic_inputs(eng, D, H, W, fsf, 5)
        mask = np.ones((H, W))
        eng.set_data(data, var, mask=mask)
        parts = [(0, (0, 20, 0, W)), (1, (31, H, 0, W))]
        eng.set_parts([r for _, r in parts], [ph for ph, _ in parts])
        assert eng.mh_layers() == 2
        eng.set_params(init)
        ra = float(max_b[0] ** 2)
        eng.mh_config(min_b, max_b, 0.1, ra, seed=11, refresh_every=0)
        err0 = eng.residual()
        accepted = eng.mh_sweeps(1, 1)
        st = O.MHState(data, var, mask, fsf, lsf, init, min_b, max_b, 0.1, ra, 11, err=err0)
        for ph in (0, 1):
            for (y, x) in part_order(parts, ph, mask, fs, fs):
                O.mh_update(st, y, x, 1)
        np.testing.assert_allclose(eng.get_params(), st.params, rtol=1e-9, atol=1e-9)
        assert accepted == st.accepted
        err = eng.download_slot(_lib.SLOT_ERR)
        assert np.max(np.abs(err - st.err)) <= 1e-11 * np.max(np.abs(st.err))
