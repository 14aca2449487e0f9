# coding=utf-8
"""
deconv3d_amd -- MI355X-native likelihood path of irap-omp/deconv3d behind the
reference's own ``Run / Instrument / SpreadFunction / LineModel`` API
(reference facade: __init__.py:10-14).
"""
from .cube import Axis, Cube, HyperspectralCube  # noqa: F401
from .instruments import MUSE, Instrument  # noqa: F401
from .line_models import LineModel, SingleGaussianLineModel  # noqa: F401
from .masks import above_percentile  # noqa: F401
from .math_utils import median_clip  # noqa: F401
from .run import Run, logger  # noqa: F401
from .spread_functions import (  # noqa: F401
    FieldSpreadFunction, GaussianFieldSpreadFunction, GaussianLineSpreadFunction,
    ImageFieldSpreadFunction, LineSpreadFunction, MoffatFieldSpreadFunction,
    MUSELineSpreadFunction, NoFieldSpreadFunction, VectorLineSpreadFunction)

__version__ = "0.1.0"
