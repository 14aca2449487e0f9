# coding=utf-8
"""Mask helpers (reference: lib/masks.py)."""
import numpy as np

from .cube import Cube


def read_hyperspectral_cube(cube):
    """lib/masks.py:6-14."""
    if isinstance(cube, str):
        cube = Cube.from_fits(cube)
    if not isinstance(cube, Cube):
        raise TypeError("Provided cube is not a HyperspectralCube")
    if cube.is_empty():
        raise ValueError("Provided cube is empty")
    return cube


def above_percentile(cube, percentile=30):
    """Mask (1/0 image) of the spaxels whose spectrally summed flux is at or
    above the given percentile (lib/masks.py:17-29)."""
    cube = read_hyperspectral_cube(cube)
    img = np.nansum(cube.data, axis=0)
    p = np.nanpercentile(img, percentile)
    return np.where(img >= p, 1.0, 0.0)
