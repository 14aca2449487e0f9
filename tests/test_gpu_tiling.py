"""
GPU test of the spatial tiling on ONE device (loop-back: every tile is its own
device context on the same GPU): the tiled chain -- global colour classes,
Philox keyed by global spaxel index, border updates replayed from 8-double
records by k_apply_updates -- is BIT-IDENTICAL to the single-context chain.
"""
import numpy as np
import pytest

from deconv3d_amd import _lib, tiling
from tests.cases import make_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,grid", [("c1", (2, 2)), ("c1", (1, 2)), ("odd_depth", (2, 1)),
                                       ("moffat", (2, 2)), ("rect_fsf", (3, 2)), ("nolsf", (2, 2))])
def test_tiled_chain_is_bit_identical_to_single_context(name, grid):
    case = make_case(name)
    D, H, W = case["D"], case["H"], case["W"]
    fh, fw = case["fsf"].shape
    ra, seed, sweeps = 35.0, 77, 2
    with _lib.Engine((D, H, W), (fh, fw)) as ref:
        ref.set_taps(case["fsf"], case["lsf"])
        ref.set_data(case["data"], case["var"], mask=case["mask"])
        ref.set_params(case["init"])
        ref.mh_config(case["min_b"], case["max_b"], 0.1, ra, seed=seed, refresh_every=0)
        accepted = ref.mh_sweeps(sweeps, 1)
        ref_params = ref.get_params()
        ref_err = ref.download_slot(_lib.SLOT_ERR)
    assert accepted > 0
    lay = tiling.TileLayout(H, W, fh, fw, *grid)
    engines = [tiling.make_tile_engine(lay, r, case["data"], case["var"], case["mask"], case["fsf"],
                                       case["lsf"], case["init"], case["min_b"], case["max_b"], 0.1,
                                       ra, seed) for r in range(lay.n)]
    try:
        steppers = [tiling.TileStepper(lay, r, engines[r], case["mask"]) for r in range(lay.n)]
        for s in range(1, sweeps + 1):
            tiling.sweep_loopback(steppers, s, fh * fw)
        live = case["mask"] == 1
        for r in range(lay.n):
            (y0, y1, x0, x1), p = tiling.gather_params(lay, r, engines[r])
            m = live[y0:y1, x0:x1]
            np.testing.assert_array_equal(p[m], ref_params[y0:y1, x0:x1][m])
            ry0, ry1, rx0, rx1 = lay.region(r)
            np.testing.assert_array_equal(engines[r].download_slot(_lib.SLOT_ERR),
                                          ref_err[:, ry0:ry1, rx0:rx1])
    finally:
        for e in engines:
            e.close()


def test_colour_counts_respect_ownership():
    case = make_case("c1")
    fh, fw = case["fsf"].shape
    lay = tiling.TileLayout(case["H"], case["W"], fh, fw, 2, 2)
    total = np.zeros(fh * fw, int)
    engines = [tiling.make_tile_engine(lay, r, case["data"], case["var"], case["mask"], case["fsf"],
                                       case["lsf"], case["init"], case["min_b"], case["max_b"], 0.1,
                                       10.0, 1) for r in range(lay.n)]
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
