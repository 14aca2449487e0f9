# coding=utf-8
"""
Field / line spread function plugins -- same class names, constructor
arguments and ``as_image`` / ``as_vector`` contracts as the reference's
``lib/spread_functions.py``.  They produce the taps once per run on the host;
the device only ever sees the resulting ``(fh, fw)`` image and ``[D]`` vector.

Pixel-unit builders (``gaussian_image`` ...) hold the arithmetic; the classes
only convert arcsec / micron to pixels through the cube's axis steps.
"""
import math

import numpy as np
from scipy.special import erf

# --------------------------------------------------------------------------- #
# pixel-unit tap builders                                                      #
# --------------------------------------------------------------------------- #


def _center(n):
    # lib/spread_functions.py:107-109, 251
    return (n - 1) // 2 - (n % 2 - 1)


def elliptical_radius(shape, pa=0., ba=1.0, xo=None, yo=None):
    """Radii of lib/spread_functions.py:113-131 on a grid of ``shape``."""
    if xo is None:
        xo = _center(shape[1])
    if yo is None:
        yo = _center(shape[0])
    y, x = np.indices(shape)
    dx = xo - x
    dy = yo - y
    t = np.radians(pa)
    dx_p = dx * np.cos(t) - dy * np.sin(t)
    dy_p = dx * np.sin(t) + dy * np.cos(t)
    return np.sqrt(dx_p ** 2 + dy_p ** 2 / ba ** 2)


def gaussian_image(fwhm_px, pa=0., ba=1.0, xo=None, yo=None):
    """Gaussian FSF, size ceil(6 sigma) made odd (lib/spread_functions.py:94-111)."""
    stddev = fwhm_px / (2 * math.sqrt(2 * math.log(2)))
    size = int(math.ceil(6. * stddev))
    if size % 2 == 0:
        size += 1
    r = elliptical_radius((size, size), pa, ba, xo, yo)
    fsf = np.exp(-0.5 * (r / stddev) ** 2)
    return fsf / fsf.sum()


def moffat_image(shape, beta, fwhm_px=None, alpha_px=None, pa=0., ba=1.0, xo=None, yo=None):
    """Moffat FSF ``(1 + (r/alpha)^2)^-beta`` on a grid of ``shape``
    (lib/spread_functions.py:165-189)."""
    r = elliptical_radius(shape, pa, ba, xo, yo)
    if alpha_px is None:
        alpha_px = fwhm_px / (2. * np.sqrt(2. ** (1. / beta) - 1))
    psf = (1. + (r / alpha_px) ** 2) ** (-beta)
    return psf / psf.sum()


def gaussian_lsf_vector_px(depth, sigma_px):
    """Gaussian LSF of length ``depth`` (lib/spread_functions.py:245-261)."""
    zc = _center(depth)
    if sigma_px == 0:
        lsf = np.zeros(depth)
        lsf[zc] = 1.
    else:
        z = np.arange(depth) - zc
        lsf = np.exp(z ** 2 / (-2. * sigma_px ** 2))
    return lsf / lsf.sum()


def muse_like_lsf_vector(depth, sigma_px=0.9, box_px=1.0):
    """
    Analytic stand-in for the MUSE LSF: a slit (box) of ``box_px`` pixels
    convolved with a Gaussian, integrated over the pixel, centred like
    lib/spread_functions.py:251, normalised to sum 1.  NOT mpdaf's qsim_v1
    model (mpdaf is not available; parity with it is unpinned).
    """
    zc = _center(depth)
    z = np.arange(depth) - zc
    s = sigma_px * math.sqrt(2.0)
    h = box_px / 2.0
    prof = 0.5 * (erf((z + h) / s) - erf((z - h) / s))
    prof[np.abs(prof) < 1e-300] = 0.0
    return prof / prof.sum()


# --------------------------------------------------------------------------- #
# field spread functions                                                       #
# --------------------------------------------------------------------------- #


class FieldSpreadFunction:
    """Interface (lib/spread_functions.py:23-36)."""

    def as_image(self, for_cube):
        """2-D image of the FSF for ``for_cube``; odd dimensions, sum 1."""
        raise NotImplementedError()


class NoFieldSpreadFunction(FieldSpreadFunction):
    """All ones of the cube's spatial shape (lib/spread_functions.py:39-52).
    Note this is the reference's behaviour, not an identity kernel; use
    ``ImageFieldSpreadFunction([[1.]])`` for no spatial spreading."""

    def __init__(self):
        pass

    def as_image(self, for_cube):
        return np.ones(for_cube.shape[1:])


class ImageFieldSpreadFunction(FieldSpreadFunction):
    """A user-provided 2-D image, used as is (lib/spread_functions.py:55-68)."""

    def __init__(self, image_2d):
        self.image_2d = image_2d

    def as_image(self, for_cube):
        return self.image_2d

    def __str__(self):
        return """Custom Image PSF"""


class GaussianFieldSpreadFunction(FieldSpreadFunction):
    """
    fwhm [arcsec], pa [deg, clockwise from Y], ba [axis ratio]
    (lib/spread_functions.py:71-131).
    """

    def __init__(self, fwhm=None, pa=0, ba=1.0):
        self.fwhm = fwhm
        self.pa = pa
        self.ba = ba

    def __str__(self):
        return "Gaussian PSF :\n    fwhm = %s \"\n    pa   = %s °\n    ba   = %s" % (
            self.fwhm, self.pa, self.ba)

    def _pixels(self, arcsec, for_cube):
        return arcsec / for_cube.get_step(1).to('arcsec').value

    def as_image(self, for_cube, xo=None, yo=None):
        return gaussian_image(self._pixels(self.fwhm, for_cube), self.pa, self.ba, xo, yo)


class MoffatFieldSpreadFunction(GaussianFieldSpreadFunction):
    """
    Moffat FSF (lib/spread_functions.py:134-189): give ``fwhm`` or ``alpha``
    (arcsec) and ``beta``.  As in the reference the image has the cube's full
    spatial shape (so even-sized cubes are rejected by Run); the optional
    ``size=`` extension evaluates it on an odd ``size x size`` grid instead
    ("Moffat PSF radius 5" = ``size=11``).
    """

    def __init__(self, fwhm=None, alpha=None, beta=None, pa=None, ba=None, size=None):
        self.alpha = alpha
        self.beta = beta
        self.size = size
        GaussianFieldSpreadFunction.__init__(self, fwhm, 0. if pa is None else pa,
                                             1.0 if ba is None else ba)

    def __str__(self):
        return "Moffat PSF :\n  fwhm  = %s \"\n  alpha = %s \"\n  beta  = %s\n  pa = %s °\n  ba = %s" % (
            self.fwhm, self.alpha, self.beta, self.pa, self.ba)

    def as_image(self, for_cube, xo=None, yo=None):
        shape = for_cube.shape[1:] if self.size is None else (self.size, self.size)
        if self.alpha is None:
            return moffat_image(shape, self.beta, fwhm_px=self._pixels(self.fwhm, for_cube),
                                pa=self.pa, ba=self.ba, xo=xo, yo=yo)
        return moffat_image(shape, self.beta, alpha_px=self._pixels(self.alpha, for_cube),
                            pa=self.pa, ba=self.ba, xo=xo, yo=yo)


# --------------------------------------------------------------------------- #
# line spread functions                                                        #
# --------------------------------------------------------------------------- #


class LineSpreadFunction:
    """Interface (lib/spread_functions.py:195-209)."""

    def as_vector(self, for_cube):
        """1-D vector of the cube's spectral length, centred at
        ``(D-1)//2 - (D%2 - 1)``, sum 1."""
        raise NotImplementedError()


class VectorLineSpreadFunction(LineSpreadFunction):
    """A user-provided vector (lib/spread_functions.py:212-228)."""

    def __init__(self, vector):
        self.vector = vector

    def as_vector(self, for_cube):
        return self.vector

    def __str__(self):
        return """Custom Vector LSF"""


class GaussianLineSpreadFunction(LineSpreadFunction):
    """Gaussian LSF, ``fwhm`` in microns (lib/spread_functions.py:231-277)."""

    def __init__(self, fwhm):
        self.fwhm = fwhm

    def __str__(self):
        return "Gaussian LSF : fwhm = %s µm \n" % self.fwhm

    def as_vector(self, for_cube):
        sigma = self.fwhm / 2.35482 / for_cube.get_step(0).to('um').value
        return gaussian_lsf_vector_px(for_cube.shape[0], sigma)

    @staticmethod
    def gaussian(x, mu, sigma):
        return np.exp((x - mu) ** 2 / (-2. * sigma ** 2))


class MUSELineSpreadFunction(LineSpreadFunction):
    """
    The reference delegates to ``mpdaf.MUSE.LSF(type=model).get_LSF``
    (lib/spread_functions.py:280-315).  When mpdaf is importable the same call
    is made; otherwise ``model='analytic'`` selects the documented stand-in
    (:func:`muse_like_lsf_vector`) and any other model raises ImportError like
    the reference does.
    """

    def __init__(self, model="qsim_v1", sigma_px=0.9, box_px=1.0):
        self.model = model
        self.sigma_px = sigma_px
        self.box_px = box_px
        self.lsf = None
        if model != "analytic":
            try:
                from mpdaf.MUSE import LSF
            except ImportError:
                raise ImportError("You need to install the mpdaf module to use "
                                  "MUSELineSpreadFunction (or pass model='analytic').")
            self.lsf = LSF(type=self.model)

    def __str__(self):
        return "MUSE LSF : model = '%s'" % self.model

    def as_vector(self, cube):
        depth = cube.shape[0]
        if self.lsf is None:
            return muse_like_lsf_vector(depth, self.sigma_px, self.box_px)
        odd_depth = depth if depth % 2 == 1 else depth + 1
        lsf_1d = self.lsf.get_LSF(lbda=cube.z_central * 1e4, step=cube.z_step * 1e4,
                                  size=odd_depth)
        if depth % 2 == 0:
            lsf_1d = lsf_1d[:-1]
        return lsf_1d / lsf_1d.sum()
