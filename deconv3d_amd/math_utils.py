# coding=utf-8
"""Host-side setup helpers (reference: lib/math_utils.py)."""
import numpy as np


def merge_where_nan(target, filler):
    """Replace the NaNs of ``target`` by ``filler``'s values, in place
    (lib/math_utils.py:4-13)."""
    np.copyto(target, filler, where=np.isnan(target))


def median_clip(data, clip_sigma=3., limit_ratio=1e-3, max_iterations=5):
    """
    Iteratively sigma-clipped median (lib/math_utils.py:16-57).
    Returns ``(median, sigma, iterations)``.
    """
    values = data[np.isfinite(data)]
    median = np.median(values)
    iteration = 0
    while True:
        iteration += 1
        previous = median
        median = np.median(values)
        sigma = np.std(values)
        kept = np.nonzero(np.abs(values - median) < clip_sigma * sigma)
        if np.size(kept) > 0:
            values = values[kept]
        if abs(median - previous) / abs(previous) < limit_ratio or iteration >= max_iterations:
            break
    return np.median(values), np.std(values), iteration
