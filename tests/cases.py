"""Seeded parity cases shared by the CPU and GPU tests (sizes the oracle
finishes in seconds).  Built with the oracle: test infrastructure only."""
import numpy as np

from oracle import deconv3d_oracle as O


def make_case(name):
    rng = np.random.default_rng(20240 + sum(map(ord, name)))
    if name == "c1":            # BASELINE config 1: 32x16x16, Gaussian 9x9, Gaussian LSF
        D, H, W = 32, 16, 16
        fsf = O.gaussian_fsf_image(3.0)
        lsf = O.gaussian_lsf_vector(D, 0.9088)
    elif name == "odd_depth":   # non power-of-two depth: partial-wrap branch of convolve_1d
        D, H, W = 21, 12, 10
        fsf = O.gaussian_fsf_image(2.2)
        lsf = O.gaussian_lsf_vector(D, 1.3)
    elif name == "d30":
        D, H, W = 30, 9, 11
        fsf = O.gaussian_fsf_image(2.5)
        lsf = O.gaussian_lsf_vector(D, 0.7)
    elif name == "asym":        # asymmetric FSF pins the convolution orientation
        D, H, W = 16, 13, 14
        fsf = O.gaussian_fsf_image(3.0, pa=30., ba=0.7)
        lsf = rng.random(D)      # dense, asymmetric, full-length LSF
        lsf /= lsf.sum()
    elif name == "moffat":      # config-2 geometry, sub-sampled: Moffat 11x11 crop
        D, H, W = 64, 20, 18
        fsf = O.moffat_cropped(11, 3.0, 2.5)
        lsf = O.gaussian_lsf_vector(D, 0.9088)
    elif name == "nolsf":       # lsf=None branch (lib/run.py:675-676)
        D, H, W = 24, 8, 9
        fsf = O.gaussian_fsf_image(2.0)
        lsf = None
    elif name == "rect_fsf":    # non-square FSF, wider than tall
        D, H, W = 16, 10, 12
        fsf = rng.random((3, 7))
        fsf /= fsf.sum()
        lsf = O.gaussian_lsf_vector(D, 0.5)
    elif name == "big_fsf":     # FSF wider than the register-tiled kernels (generic path)
        D, H, W = 8, 21, 23
        fsf = O.moffat_fsf_image((21, 23), 2.5, fwhm_px=3.0)   # cube-sized, as the reference's Moffat
        lsf = O.gaussian_lsf_vector(D, 0.6)
    elif name == "tiny":        # 1 spaxel wide, D=2
        D, H, W = 2, 1, 3
        fsf = O.gaussian_fsf_image(1.0)
        lsf = O.gaussian_lsf_vector(D, 0.4)
    elif name == "tile_a":      # tiling: 5x7 asymmetric FSF, tiles >= 8 rows x 12 columns
        D, H, W = 12, 34, 26
        fsf = rng.random((5, 7))
        fsf /= fsf.sum()
        lsf = O.gaussian_lsf_vector(D, 0.8)
    elif name == "tile_b":      # tiling: tall 9x3 FSF, tiles >= 16 rows x 4 columns
        D, H, W = 16, 40, 18
        fsf = np.outer(O.gaussian_fsf_image(3.0)[:, 4], [0.25, 0.5, 0.25])
        fsf /= fsf.sum()
        lsf = O.gaussian_lsf_vector(D, 0.9088)
    elif name == "tile_deep":   # tiling a cube beyond 512 channels: the z-blocked sweep kernels
        D, H, W = 600, 26, 26
        fsf = O.gaussian_fsf_image(2.0)                         # 7x7
        lsf = O.muse_like_lsf(D)                                # taps within +-8 channels
    else:
        raise KeyError(name)
    y, x = np.indices((H, W))
    truth = np.dstack((
        1.0 + 9.0 * rng.random((H, W)),
        D * (0.25 + 0.5 * rng.random((H, W))),
        0.8 + 2.0 * rng.random((H, W)),
    ))
    mask = np.ones((H, W))
    if H * W > 20:
        mask[rng.integers(0, H, 3), rng.integers(0, W, 3)] = 0
    clean = O.forward_full((D, H, W), truth, mask, fsf, lsf)
    sigma = 0.05 * np.max(clean) + 1e-3
    data = clean + rng.normal(0., sigma, size=(D, H, W))
    var = (sigma * (0.5 + rng.random((D, H, W)))) ** 2      # non-uniform variance
    min_b = O.model_min_boundaries()
    max_b = O.model_max_boundaries(data, fsf)
    init = min_b + (max_b - min_b) * rng.random((H, W, 3))
    init[..., 2] = np.maximum(init[..., 2], 0.3)
    return dict(name=name, D=D, H=H, W=W, fsf=fsf, lsf=lsf, truth=truth, mask=mask,
                data=data, var=var, min_b=min_b, max_b=max_b, init=init, rng=rng)


ALL_CASES = ["c1", "odd_depth", "d30", "asym", "moffat", "nolsf", "rect_fsf", "big_fsf", "tiny"]
