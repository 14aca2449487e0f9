#!/usr/bin/env python
# coding=utf-8
"""
Generates the committed fixtures under tests/golden/.  Run in the BUILD
container only (it imports the importable pieces of the reference from
/root/reference/lib and reads the reference's own Matlab fixtures); the GPU box
never sees /root/reference, only the .npz files written here.

    python tests/golden/make_goldens.py

Fixtures
  ref_line_model.npz   outputs of the reference's SingleGaussianLineModel
                       (lib/line_models.py) -- pins oracle.gaussian_line/bounds
  ref_median_clip.npz  outputs of the reference's median_clip (lib/math_utils.py)
  ref_rtnorm.npz       sorted draws of the reference's rtnorm (lib/rtnorm.py),
                       seeded numpy global RNG -- distribution pin for the own
                       truncated-normal sampler (KS tests)
  ref_mat_fixture.npz  the reference's own data fixture
                       (tests/input/data14forAntoine.mat, Parametres_theoriques.mat)
                       re-laid out as (D,H,W) / (H,W,3): data, variance, 15x15
                       FSF, theoretical parameters (c already 0-based)
  ref_galpak_pair.npz  the reference's saved convolved/deconvolved cube pair
                       (tests/input/GalPaK_*_myrun100k_*.fits)
  oracle_*.npz         oracle outputs on the seeded parity cases (regression
                       pins + the vectors the GPU tests compare against)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "lib"))

from oracle import deconv3d_oracle as O  # noqa: E402
from tests.cases import make_case  # noqa: E402


def ref_line_model():
    import line_models as ref
    m = ref.SingleGaussianLineModel()
    rng = np.random.default_rng(1)
    x = np.arange(64, dtype=np.float64)
    params = np.column_stack((rng.random(32) * 10, rng.random(32) * 63, 0.2 + rng.random(32) * 8))
    lines = np.array([m.modelize(None, x, p) for p in params])

    class FakeCube:
        data = rng.random((12, 5, 6)) * 7.0

    class FakeRunner:
        cube = FakeCube()
        fsf = O.gaussian_fsf_image(3.0)

    np.savez_compressed(os.path.join(HERE, "ref_line_model.npz"), x=x, params=params, lines=lines,
                        names=np.array(m.parameters()), gibbs_index=m.gibbs_parameter_index(),
                        min_b=np.array(m.min_boundaries(FakeRunner()), dtype=np.float64),
                        max_b=np.array(m.max_boundaries(FakeRunner()), dtype=np.float64),
                        cube=FakeCube.data, fsf=FakeRunner.fsf)


def ref_median_clip():
    import math_utils as ref
    rng = np.random.default_rng(2)
    data = rng.normal(3.0, 2.0, size=(26, 24, 2))
    data[3, 4, 1] = 80.0
    data[9, 1, 0] = -55.0
    data[0, 0, 0] = np.nan
    med, sig, it = ref.median_clip(data.copy(), 2.5)
    np.savez_compressed(os.path.join(HERE, "ref_median_clip.npz"), data=data, clip=2.5,
                        median=med, sigma=sig, iterations=it)


RTNORM_CASES = [  # (a, b, mu, sigma)
    (0.0, 50.0, 1.0, 1.0),        # body, Chopin tables
    (0.0, 50.0, -1.5, 1.0),       # right tail, lower bound at 1.5 sigma
    (0.0, 50.0, -8.0, 1.0),       # far tail (exponential rejection in both samplers)
    (0.0, 9.0, 4.0, 3.0),         # two-sided, wide
    (0.0, 0.5, 2.0, 1.5),         # narrow interval
    (-1.0, 2.0, 0.0, 1.0),        # straddles the mode
    (0.0, 30.0, 45.0, 4.0),       # mass against the upper bound
    # LAST: a regime where the reference's python port of Chopin's sampler is
    # itself biased (lower bound 2.3-3.1 sigma, far upper bound: mean 0.263 vs
    # the truncated normal's 0.290 at 2.9 sigma).  Kept to document the quirk;
    # the own sampler follows the analytic distribution there.
    (0.0, 50.0, -2.9, 1.0),
]


def ref_rtnorm():
    import rtnorm as ref
    np.random.seed(20150625)
    n = 6000
    out = {}
    for i, (a, b, mu, sg) in enumerate(RTNORM_CASES):
        out["draws_%d" % i] = np.sort(ref.rtnorm(a, b, mu=mu, sigma=sg, size=n))
    np.savez_compressed(os.path.join(HERE, "ref_rtnorm.npz"), cases=np.array(RTNORM_CASES), **out)


def ref_mat_fixture():
    from scipy.io import loadmat
    mat = loadmat(os.path.join(REF, "tests/input/data14forAntoine.mat"))
    th = loadmat(os.path.join(REF, "tests/input/Parametres_theoriques.mat"))["Parametres_theoriques"]
    # tests/read_mat.py:33-36, 52-65: transpose to (Z,Y,X); Matlab c is 1-based
    data = np.ascontiguousarray(np.transpose(mat["data_noise"]))
    var = np.ascontiguousarray(np.transpose(mat["varNoise"]))
    fsf = np.ascontiguousarray(np.transpose(mat["FSF"]))
    a = np.transpose(th[:, :, 0])
    c = np.transpose(th[:, :, 1]) - 1.0
    w = np.transpose(th[:, :, 2])
    params = np.ascontiguousarray(np.dstack((a, c, w)))
    np.savez_compressed(os.path.join(HERE, "ref_mat_fixture.npz"), data=data, var=var, fsf=fsf,
                        params=params)


def ref_galpak_pair():
    """The reference's own saved outputs of a 100k-iteration MUSE run
    (tests/input/GalPaK_..._myrun100k_{convolved,deconvolved}_cube.fits, written
    by Run.save_fits, lib/run.py:797-806): convolved = LSF x FSF of deconvolved
    under the MUSE defaults (lib/instruments.py:95-107) to 0.4 % of the peak."""
    from deconv3d_amd.cube import Cube
    stem = os.path.join(REF, "tests/input/GalPaK_cube_1101_size4.08_flux1e-16_incl60_vmax199_"
                             "disp80_seeing1.00_myrun100k_")
    conv = Cube.from_fits(stem + "convolved_cube.fits").data
    clean = Cube.from_fits(stem + "deconvolved_cube.fits").data
    np.savez_compressed(os.path.join(HERE, "ref_galpak_pair.npz"), convolved=conv, clean=clean)


def oracle_cases():
    for name in ("c1", "odd_depth", "asym"):
        case = make_case(name)
        shape = (case["D"], case["H"], case["W"])
        clean = O.simulate_clean(shape, case["truth"], case["mask"])
        lines = O.lsf_lines(shape, case["truth"], case["mask"], case["lsf"])
        sim = O.forward_full(shape, case["truth"], case["mask"], case["fsf"], case["lsf"])
        err = O.compute_error_in_one_step(case["data"], case["init"], case["mask"], case["fsf"],
                                          case["lsf"])
        cmap = O.chi2_map(err, case["var"])
        rng = np.random.default_rng(99)
        probes = []
        for _ in range(32):
            y, x = int(rng.integers(0, case["H"])), int(rng.integers(0, case["W"]))
            p_new = case["init"][y, x] + np.array([0., 0.8, 0.25]) * rng.normal(size=3)
            p_new[2] = abs(p_new[2]) + 0.1
            st = O.window_stats(err, case["var"], case["init"][y, x], p_new, y, x, case["fsf"],
                                case["lsf"])
            probes.append(np.concatenate(([y, x], p_new, st)))
        # a short deterministic chain (Philox seed 2024, colour order)
        stt = O.MHState(case["data"], case["var"], case["mask"], case["fsf"], case["lsf"],
                        case["init"], case["min_b"], case["max_b"], seed=2024)
        chain = [stt.params.copy()]
        dlog = []
        for s in (1, 2):
            O.mh_sweep(stt, s)
            chain.append(stt.params.copy())
            dlog.append(stt.dlog.copy())
        np.savez_compressed(
            os.path.join(HERE, "oracle_%s.npz" % name),
            fsf=case["fsf"], lsf=case["lsf"], truth=case["truth"], init=case["init"],
            mask=case["mask"], data=case["data"], var=case["var"], min_b=case["min_b"],
            max_b=case["max_b"], clean=clean, lsf_lines=lines, sim=sim, err=err, chi2_map=cmap,
            probes=np.array(probes), chain=np.array(chain), dlog=np.array(dlog),
            accepted=stt.accepted, ra=stt.ra, chain_seed=2024)


if __name__ == "__main__":
    ref_line_model()
    ref_median_clip()
    ref_rtnorm()
    ref_mat_fixture()
    ref_galpak_pair()
    oracle_cases()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print("%-28s %8d bytes" % (f, os.path.getsize(os.path.join(HERE, f))))
