"""
CPU tests of the spatial tiling, no GPU:
  * layout geometry (ownership partition, who reports what to whom),
  * the per-colour protocol driven through an oracle-backed engine: a tiled
    chain is BIT-IDENTICAL to the single-domain chain, both with all tiles in
    one process (loop-back) and as a world_size-2 torch.distributed job (gloo).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from deconv3d_amd import tiling
from oracle import deconv3d_oracle as O
from tests.cases import make_case
from tests.tiling_oracle import OracleTileEngine

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_layout_partitions_the_grid():
    lay = tiling.TileLayout(30, 23, 7, 5, 2, 3)
    cover = np.zeros((30, 23), int)
    for r in range(lay.n):
        y0, y1, x0, x1 = lay.owned(r)
        cover[y0:y1, x0:x1] += 1
        ry0, ry1, rx0, rx1 = lay.region(r)
        assert ry0 == max(y0 - 3, 0) and ry1 == min(y1 + 3, 30)
        assert rx0 == max(x0 - 2, 0) and rx1 == min(x1 + 2, 23)
    assert (cover == 1).all()
    assert tiling.tile_grid_for(2) == (1, 2) and tiling.tile_grid_for(4) == (2, 2)
    assert tiling.tile_grid_for(8) == (2, 4) and tiling.tile_grid_for(1) == (1, 1)
    with pytest.raises(ValueError):
        tiling.TileLayout(4, 4, 3, 3, 5, 1)


def test_send_lists_cover_every_window_that_reaches_a_neighbour():
    H, W, fh, fw = 26, 21, 5, 7
    lay = tiling.TileLayout(H, W, fh, fw, 2, 2)
    mask = np.ones((H, W))
    mask[3, 4] = 0
    for r in range(lay.n):
        lists = lay.send_lists(r, mask)
        y0, y1, x0, x1 = lay.owned(r)
        for nb in range(lay.n):
            if nb == r:
                continue
            ry0, ry1, rx0, rx1 = lay.region(nb)
            need = set()
            for y in range(y0, y1):
                for x in range(x0, x1):
                    if mask[y, x] != 1:
                        continue
                    # window of (y,x) intersects nb's stored region?
                    if y + lay.fhh >= ry0 and y - lay.fhh < ry1 and \
                            x + lay.fhw >= rx0 and x - lay.fhw < rx1:
                        need.add((y, x))
            got = set()
            for c, yx in enumerate(lists.get(nb, [[]] * (fh * fw))):
                for (y, x) in yx:
                    assert (y % fh) * fw + (x % fw) == c
                    got.add((int(y), int(x)))
            assert need <= got, (r, nb, sorted(need - got)[:5])


def run_single(case, sweeps, seed):
    st = O.MHState(case["data"], case["var"], case["mask"], case["fsf"], case["lsf"],
                   case["init"], case["min_b"], case["max_b"], 0.1, 40.0, seed)
    for s in range(1, sweeps + 1):
        O.mh_sweep(st, s)
    return st


@pytest.mark.parametrize("name,grid", [("c1", (2, 2)), ("odd_depth", (1, 2)), ("rect_fsf", (3, 2))])
def test_loopback_tiled_chain_is_bit_identical(name, grid):
    case = make_case(name)
    H, W = case["H"], case["W"]
    fh, fw = case["fsf"].shape
    lay = tiling.TileLayout(H, W, fh, fw, *grid)
    err0 = O.compute_error_in_one_step(case["data"], case["init"], case["mask"], case["fsf"],
                                       case["lsf"])
    engines = [OracleTileEngine(lay, r, case["data"], case["var"], case["mask"], case["fsf"],
                                case["lsf"], case["init"], case["min_b"], case["max_b"], 0.1, 40.0,
                                9, err0) for r in range(lay.n)]
    steppers = [tiling.TileStepper(lay, r, engines[r], case["mask"]) for r in range(lay.n)]
    for s in (1, 2):
        tiling.sweep_loopback(steppers, s, fh * fw)
    ref = run_single(case, 2, 9)
    for r in range(lay.n):
        (y0, y1, x0, x1), p = tiling.gather_params(lay, r, engines[r])
        np.testing.assert_array_equal(p, ref.params[y0:y1, x0:x1])
        ry0, ry1, rx0, rx1 = lay.region(r)
        np.testing.assert_array_equal(engines[r].st.err, ref.err[:, ry0:ry1, rx0:rx1])


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
case = make_case("c1")
fh, fw = case["fsf"].shape
lay = tiling.TileLayout(case["H"], case["W"], fh, fw, *tiling.tile_grid_for(world))
err0 = O.compute_error_in_one_step(case["data"], case["init"], case["mask"], case["fsf"], case["lsf"])
eng = OracleTileEngine(lay, rank, case["data"], case["var"], case["mask"], case["fsf"], case["lsf"],
                       case["init"], case["min_b"], case["max_b"], 0.1, 40.0, 9, err0)
st = tiling.TileStepper(lay, rank, eng, case["mask"])
for s in (1, 2):
    tiling.sweep_distributed(st, s, fh * fw, dist, torch, None)
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
    ref = run_single(make_case("c1"), 2, 9)
    for r in range(2):
        y0, y1, x0, x1 = np.load(tmp_path / ("rect_%d.npy" % r))
        np.testing.assert_array_equal(np.load(tmp_path / ("params_%d.npy" % r)),
                                      ref.params[y0:y1, x0:x1])
