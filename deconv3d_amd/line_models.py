# coding=utf-8
"""
Line-model plugin interface -- same names and contracts as the reference's
``lib/line_models.py`` (LineModel :4-61, SingleGaussianLineModel :64-109).

The device kernels implement ``SingleGaussianLineModel`` (amplitude Gibbs-
sampled, centre/width Metropolis-Hastings).  ``modelize`` below is the host
evaluation of the same curve, kept for API parity (plots, user scripts).
"""
import numpy as np


class LineModel:
    """Interface of a spectral line model (lib/line_models.py:4-61)."""

    def __init__(self):
        pass

    def parameters(self):
        """Names of the parameters (unique strings)."""
        raise NotImplementedError()

    def gibbs_parameter_index(self):
        """Index of the Gibbs-sampled parameter (the amplitude), or None."""
        return None

    def min_boundaries(self, runner):
        raise NotImplementedError()

    def max_boundaries(self, runner):
        raise NotImplementedError()

    def post_jump(self, runner, old_parameters, new_parameters):
        """Optional hook mutating ``new_parameters`` after the Cauchy jump."""
        pass

    def modelize(self, runner, x, parameters):
        raise NotImplementedError()


class SingleGaussianLineModel(LineModel):
    """``a * exp(-(x-c)^2 / (2 w^2))`` with parameters ``['a', 'c', 'w']``."""

    def parameters(self):
        return ['a', 'c', 'w']

    def gibbs_parameter_index(self):
        return 0

    def min_boundaries(self, runner):
        return [0, 0, 0]

    def max_boundaries(self, runner):
        # lib/line_models.py:79-90: the FSF is normalised, so the amplitude
        # ceiling is max(data)/max(fsf); centre in [0, D-1], width in [0, D].
        # NaN voxels (masked spectra, lib/run.py:157-162) are ignored: the
        # reference's np.amax would make every bound NaN.
        data = runner.cube.data
        fsf_max = np.amax(runner.fsf)
        a_max = np.nanmax(data)
        if fsf_max > 0:
            a_max = a_max / fsf_max
        return [a_max, data.shape[0] - 1, data.shape[0]]

    def modelize(self, runner, x, parameters):
        return self.gaussian(np.asarray(x, dtype=np.float64),
                             parameters[0], parameters[1], parameters[2])

    @staticmethod
    def gaussian(x, a, c, w):
        return a * np.exp(-1. * (x - c) ** 2 / (2. * w ** 2))
