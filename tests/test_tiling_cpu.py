"""
CPU tests of the spatial tiling, no GPU:
  * layout geometry: tiles partition the grid, parts partition the tiles, parts of
    one phase never touch a common cell (the property the protocol rests on), the
    halo plans of two ranks describe the same rectangles;
  * the phase / halo-copy protocol driven through an oracle-backed engine: a tiled
    chain is BIT-IDENTICAL to the single-domain chain scanned in the same part
    order, with all tiles in one process (loop-back) and as a world_size-2
    torch.distributed job (gloo); the from-scratch residual of lib/run.py:521-534
    with its parameter gather;
  * `bench.py --gpus 2 --backend gloo --dry-run`: the rank / seed plumbing of the
    self-started ensemble.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from deconv3d_amd import tiling
from oracle import deconv3d_oracle as O
from tests.cases import make_case
from tests.tiling_oracle import OracleTileEngine, sweep_in_part_order

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_layout_partitions_the_grid():
    lay = tiling.TileLayout(60, 46, 7, 5, 2, 3)
    cover = np.zeros((60, 46), int)
    for r in range(lay.n):
        y0, y1, x0, x1 = lay.owned(r)
        cover[y0:y1, x0:x1] += 1
        assert lay.used(r) == (max(y0 - 3, 0), min(y1 + 3, 60), max(x0 - 2, 0), min(x1 + 2, 46))
        assert lay.region(r) == (max(y0 - 6, 0), min(y1 + 6, 60), max(x0 - 4, 0), min(x1 + 4, 46))
        inner = np.zeros((60, 46), int)
        for ph, (py0, py1, px0, px1) in lay.parts(r):
            assert 0 <= ph < 4
            inner[py0:py1, px0:px1] += 1
        assert (inner[y0:y1, x0:x1] == 1).all() and inner.sum() == (y1 - y0) * (x1 - x0)
    assert (cover == 1).all()
    assert tiling.tile_grid_for(8) == (8, 1) and tiling.parse_tiles("2x4", 8) == (2, 4)
    with pytest.raises(ValueError):
        tiling.parse_tiles("2x2", 8)
    with pytest.raises(ValueError):
        tiling.TileLayout(4, 4, 3, 3, 5, 1)
    with pytest.raises(ValueError):          # tiles of 8 rows cannot hold a 9-row FSF's phases
        tiling.TileLayout(16, 16, 9, 9, 2, 2)


@pytest.mark.parametrize("H,W,fh,fw,ty,tx", [
    (300, 300, 11, 11, 2, 1), (300, 300, 11, 11, 8, 1), (300, 300, 11, 11, 2, 2),
    (300, 300, 11, 11, 2, 4), (34, 26, 5, 7, 4, 2), (40, 18, 9, 3, 2, 4), (50, 9, 3, 1, 5, 1),
    (33, 47, 5, 5, 1, 5)])
def test_parts_of_a_phase_never_touch_a_common_cell(H, W, fh, fw, ty, tx):
    lay = tiling.TileLayout(H, W, fh, fw, ty, tx)
    assert lay.check_disjoint()
    # row strips and column strips need two phases, 2-D grids four
    assert len(lay.phases) == (1 if ty * tx == 1 else 2 if min(ty, tx) == 1 else 4)
    # what rank r sends to r' after a phase is what r' expects from r, and it covers
    # every cell r touched that r' uses
    for ph in lay.phases:
        for r in range(lay.n):
            for peer, send, recv in lay.halo_entries(r, ph):
                back = {p: (s, rc) for p, s, rc in lay.halo_entries(peer, ph)}[r]
                assert back == (recv, send)
                t, u = lay.touched(r, ph), lay.used(peer)
                assert send == tiling._intersect(t, u)
    for r in range(lay.n):
        for peer, send, recv in lay.param_entries(r):
            back = {p: (s, rc) for p, s, rc in lay.param_entries(peer)}[r]
            assert back == (recv, send)


def run_single(case, lay, sweeps, seed):
    st = O.MHState(case["data"], case["var"], case["mask"], case["fsf"], case["lsf"],
                   case["init"], case["min_b"], case["max_b"], 0.1, 40.0, seed)
    for s in range(1, sweeps + 1):
        sweep_in_part_order(st, lay, s)
    return st


def make_engines(case, lay, seed, err0):
    return [OracleTileEngine(lay, r, case["data"], case["var"], case["mask"], case["fsf"],
                             case["lsf"], case["init"], case["min_b"], case["max_b"], 0.1, 40.0,
                             seed, err0) for r in range(lay.n)]


@pytest.mark.parametrize("name,grid", [("tile_a", (2, 2)), ("tile_a", (4, 1)), ("tile_a", (1, 2)),
                                       ("tile_b", (2, 3)), ("c1", (1, 1))])
def test_loopback_tiled_chain_is_bit_identical(name, grid):
    case = make_case(name)
    fh, fw = case["fsf"].shape
    lay = tiling.TileLayout(case["H"], case["W"], fh, fw, *grid)
    err0 = O.compute_error_in_one_step(case["data"], case["init"], case["mask"], case["fsf"],
                                       case["lsf"])
    engines = make_engines(case, lay, 9, err0)
    tables = [e.tables for e in engines]
    for s in (1, 2):
        tiling.sweep_loopback(engines, lay, tables, s)
    ref = run_single(case, lay, 2, 9)
    accepted = 0
    for r in range(lay.n):
        (y0, y1, x0, x1), p = tiling.gather_params(lay, r, engines[r])
        np.testing.assert_array_equal(p, ref.params[y0:y1, x0:x1])
        # every cell the rank uses holds the single-domain residual, bit for bit
        uy0, uy1, ux0, ux1 = lay.used(r)
        ry0, _, rx0, _ = lay.region(r)
        np.testing.assert_array_equal(
            engines[r].st.err[:, uy0 - ry0:uy1 - ry0, ux0 - rx0:ux1 - rx0],
            ref.err[:, uy0:uy1, ux0:ux1])
        accepted += engines[r].mh_accepted()
    assert accepted == ref.accepted


def test_part_order_is_a_permutation_of_the_colour_order():
    """The tiled scan order visits every unmasked spaxel exactly once per sweep."""
    case = make_case("tile_a")
    fh, fw = case["fsf"].shape
    lay = tiling.TileLayout(case["H"], case["W"], fh, fw, 2, 2)
    from tests.tiling_oracle import part_order
    seen = [yx for ph in lay.phases
            for yx in part_order(lay.all_parts(), ph, case["mask"], fh, fw)]
    assert sorted(seen) == sorted(O.colour_order(case["mask"], fh, fw))
    assert len(set(seen)) == len(seen)


def test_refresh_with_parameter_gather_rebuilds_the_used_cells():
    """lib/run.py:521-534 in a tiled run: after the parameter gather every rank's
    from-scratch residual equals the global one on the cells it uses."""
    case = make_case("tile_b")
    fh, fw = case["fsf"].shape
    lay = tiling.TileLayout(case["H"], case["W"], fh, fw, 2, 2)
    engines = make_engines(case, lay, 4, None)        # local initial residual
    tables = [e.tables for e in engines]
    tiling.sweep_loopback(engines, lay, tables, 1)
    tiling.sweep_loopback(engines, lay, tables, 2, refresh=True)
    params = np.array(case["init"])
    for r in range(lay.n):
        (y0, y1, x0, x1), p = tiling.gather_params(lay, r, engines[r])
        params[y0:y1, x0:x1] = p
    want = O.compute_error_in_one_step(case["data"], params, case["mask"], case["fsf"], case["lsf"])
    ref = run_single(case, lay, 2, 4)
    np.testing.assert_array_equal(params, ref.params)     # local start == global start on used cells
    for r in range(lay.n):
        uy0, uy1, ux0, ux1 = lay.used(r)
        ry0, _, rx0, _ = lay.region(r)
        got = engines[r].st.err[:, uy0 - ry0:uy1 - ry0, ux0 - rx0:ux1 - rx0]
        np.testing.assert_allclose(got, want[:, uy0:uy1, ux0:ux1], rtol=0, atol=1e-12)


WORKER = r"""
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
from deconv3d_amd import tiling
from oracle import deconv3d_oracle as O
from tests.cases import make_case
from tests.tiling_oracle import OracleTileEngine

dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
case = make_case("tile_a")
fh, fw = case["fsf"].shape
lay = tiling.TileLayout(case["H"], case["W"], fh, fw, *tiling.tile_grid_for(world))
err0 = O.compute_error_in_one_step(case["data"], case["init"], case["mask"], case["fsf"], case["lsf"])
eng = OracleTileEngine(lay, rank, case["data"], case["var"], case["mask"], case["fsf"], case["lsf"],
                       case["init"], case["min_b"], case["max_b"], 0.1, 40.0, 9, err0)
for s in (1, 2):
    tiling.sweep_distributed(eng, lay, eng.tables, s, dist, torch)
(y0, y1, x0, x1), p = tiling.gather_params(lay, rank, eng)
np.save(os.path.join(%(out)r, "params_%%d.npy" %% rank), p)
np.save(os.path.join(%(out)r, "rect_%%d.npy" %% rank), np.array([y0, y1, x0, x1]))
dist.barrier()
dist.destroy_process_group()
"""


def test_distributed_gloo_world2_is_bit_identical(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "out": str(tmp_path)})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    case = make_case("tile_a")
    lay = tiling.TileLayout(case["H"], case["W"], *case["fsf"].shape, *tiling.tile_grid_for(2))
    ref = run_single(case, lay, 2, 9)
    for r in range(2):
        y0, y1, x0, x1 = np.load(tmp_path / ("rect_%d.npy" % r))
        np.testing.assert_array_equal(np.load(tmp_path / ("params_%d.npy" % r)),
                                      ref.params[y0:y1, x0:x1])


def test_bench_starts_its_own_ranks_dry_run():
    """`bench.py --gpus 2 --backend gloo --dry-run` with no launcher in the
    environment: the parent starts two ranks, which rendezvous over gloo and report
    rank, world size and chain seed (12345 + rank, BASELINE config 5); rank 0
    prints one JSON line with n_gpus = 2.  No GPU is touched."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2",
                        "--backend", "gloo", "--dry-run", "--steps", "3", "--warmup", "1"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["dry_run"] is True and out["n_gpus"] == 2 and out["steps"] == 3
    assert out["ranks"] == [0, 1] and out["seeds"] == [12345, 12346]
    assert out["config"]["parallelism"] == "ensemble of 2 chains"


def test_rank_classes_and_the_critical_path_of_a_tiled_sweep():
    """a tiled sweep's compute time is the sum over the PHASES of the slowest rank in
    each (a rank without a part in a phase waits) -- not one rank's total.  rank_classes groups
    the ranks that run the same launches; critical_path adds up per-phase maxima."""
    lay = tiling.TileLayout(300, 300, 11, 11, 2, 4)
    classes = tiling.rank_classes(lay)
    assert sorted(sum(classes.values(), [])) == list(range(8))
    # ranks 0-2 own all four parts (FF, FN, NF, NN); rank 7 (last row, last column) only FF
    by_rank = {r: key for key, ranks in classes.items() for r in ranks}
    assert [ph for ph, _, _ in by_rank[0]] == [0, 1, 2, 3] and by_rank[0] == by_rank[1] == by_rank[2]
    assert [ph for ph, _, _ in by_rank[7]] == [0] and [ph for ph, _, _ in by_rank[3]] == [0, 2]
    strips = tiling.rank_classes(tiling.TileLayout(300, 300, 11, 11, 8, 1))
    assert all([ph for ph, _, _ in key] in ([0, 2], [0]) for key in strips)
    # a 2 x 2 grid: rank 0 is the busiest, but the FF phase is set by the rank with the largest FF part
    times = {0: {0: 2.0, 1: 1.3, 2: 1.3, 3: 1.1}, 1: {0: 2.05, 2: 1.32}, 2: {0: 2.03, 1: 1.31}, 3: {0: 2.1}}
    total, per_phase, busiest = tiling.critical_path(times)
    assert abs(total - (2.1 + 1.31 + 1.32 + 1.1)) < 1e-12
    assert per_phase[0] == (3, 2.1) and per_phase[1] == (2, 1.31) and per_phase[3] == (0, 1.1)
    assert busiest == 0 and total > sum(times[0].values())


def test_bench_counts_the_lsf_taps_the_library_keeps():
    """bench.kept_lsf_taps mirrors d3d_set_taps' default cut (an error bound of 1e-16 of the sum):
    the MUSE-like stand-in keeps 17 taps, a Gaussian of sigma 1.0 px fits +-8 channels, one of
    sigma 1.1 px does not -- and round 3's 1e-20 of the largest tap kept 19 of the sigma-1.0 LSF."""
    import bench
    from deconv3d_amd.spread_functions import gaussian_lsf_vector_px, muse_like_lsf_vector
    assert bench.kept_lsf_taps(muse_like_lsf_vector(128, sigma_px=0.9, box_px=1.0)) == 17
    g10 = gaussian_lsf_vector_px(128, 1.0)
    assert bench.kept_lsf_taps(g10) == 17
    assert int(np.count_nonzero(g10 > 1e-20 * g10.max())) == 19
    assert bench.kept_lsf_taps(gaussian_lsf_vector_px(128, 1.1)) > 17
