# coding=utf-8
"""
CPU ORACLE for the deconv3d likelihood hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain numpy/scipy fp64 restatement of the reference algorithm
(irap-omp/deconv3d v0.3.0, python 2 + numpy).  It is the *checker* for the HIP
kernels.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it; the product package ``deconv3d_amd`` never
does, and fails loudly when its HIP library is missing.

Every function cites the reference ``file:line`` it follows (paths relative to
the reference root).  Where the reference is python-2 only (integer ``/``,
list-of-slices indexing) the restatement uses ``//`` and tuples; nothing else
is changed.

Pinning status (see DESIGN.md "Oracle"):
  * the reference's own tests hold NO numerical golden for this path
     -> the restatement is pinned by
      - the importable reference modules ``lib/line_models.py``,
        ``lib/math_utils.py``, ``lib/rtnorm.py`` run in the build container
        (tests/golden/make_goldens.py, fixtures committed under tests/golden/),
      - ``scipy.signal.convolve2d`` (the routine the reference itself calls,
        lib/run.py:1027-1029),
      - the verbatim FFT pipeline of ``lib/convolution.py`` restated below and
        checked against the closed form used on the device,
      - the statistical known-answer of the reference's Matlab fixture
        (tests/input/data14forAntoine.mat + Parametres_theoriques.mat),
      - the reference's saved convolved/deconvolved FITS pair, reproduced to
        1e-15 through the older revision's 3-D FFT convolution restated below
        (legacy_convolve_3d_same): a bit-level pin of padding / convolve_1d /
        the Gaussian LSF vector at the non-power-of-two depth 30.
  * ``MUSELineSpreadFunction`` arithmetic lives in mpdaf (absent, unpinned
    version) -> parity unpinned for that class; treated as an input vector.

Array conventions are the reference's: cube ``(D, H, W)`` C-order, parameters
``(H, W, 3)`` = ``(a, c, w)``, FSF ``(fh, fw)`` odd, LSF ``[D]``.
"""
from __future__ import annotations

import math

import numpy as np
from scipy.signal import convolve2d
from scipy.special import erfc, erfcinv, ndtr, ndtri

# --------------------------------------------------------------------------- #
# a1 -- line model                                                             #
# --------------------------------------------------------------------------- #


def gaussian_line(x, a, c, w):
    """lib/line_models.py:98-109  ``a * exp(-(x-c)**2 / (2 w**2))``."""
    x = np.asarray(x, dtype=np.float64)
    return a * np.exp(-1. * (x - c) ** 2 / (2. * w ** 2))


def model_min_boundaries():
    """lib/line_models.py:76-77."""
    return np.array([0., 0., 0.])


def model_max_boundaries(data, fsf):
    """lib/line_models.py:79-90  [max(data)/max(fsf), D-1, D]."""
    fsf_max = np.amax(fsf)
    a_max = np.amax(data)
    if fsf_max > 0:
        a_max = a_max / fsf_max
    return np.array([a_max, data.shape[0] - 1., float(data.shape[0])])


# --------------------------------------------------------------------------- #
# a2 / a3 -- spectral convolution                                              #
# --------------------------------------------------------------------------- #


def padded_length(depth):
    """lib/convolution.py:137-141  ``2 ** len(bin(depth-1)[:-1] + '0')``."""
    s = np.binary_repr(depth - 1)
    s = s[:-1] + '0'
    return 2 ** len(s)


def padding_offset(depth):
    """lib/convolution.py:149-155  offset of the data inside the padded vector."""
    n = padded_length(depth)
    diff = n - depth
    if diff & 1:
        return diff // 2 + 1
    return diff // 2


def padding_1d(vec):
    """lib/convolution.py:123-160 restricted to one axis (the only live use)."""
    vec = np.asarray(vec, dtype=np.float64)
    depth = vec.shape[0]
    n = padded_length(depth)
    half = padding_offset(depth)
    padded = np.zeros(n)
    box = slice(half, depth + half)
    padded[box] = vec.copy()
    return padded, box


def convolve_1d_fft(line, lsf):
    """
    lib/convolution.py:89-120, verbatim pipeline:
    crop(fftshift(irfft(rfft(pad(line)) * rfft(pad(lsf))))).
    """
    cubep, boxcube = padding_1d(line)
    size = cubep.shape[0]
    psfp, _ = padding_1d(lsf)
    fftpsf = np.fft.rfftn(psfp, s=[size], axes=[0])
    fftimg = np.fft.rfftn(cubep, s=[size], axes=[0])
    fft = np.fft.fftshift(
        np.fft.irfftn(fftimg * fftpsf, s=[size], axes=[0]), axes=[0]).real
    return fft[boxcube]


def lsf_taps(lsf, rel_threshold=0.0):
    """
    Closed form of ``convolve_1d``:

        out[k] = sum_t lsf[t] * line_ext[(k + s_t) mod N],
        s_t = N/2 - h - t,  line_ext[j] = line[j] if j < D else 0,

    N = padded_length(D), h = padding_offset(D).  Returns ``(shifts, weights,
    N)`` for the taps with ``|lsf[t]| > rel_threshold * max|lsf|`` (0 keeps
    every non-zero tap).  This is the form the device kernels evaluate.
    """
    lsf = np.asarray(lsf, dtype=np.float64)
    depth = lsf.shape[0]
    n = padded_length(depth)
    h = padding_offset(depth)
    thr = rel_threshold * np.max(np.abs(lsf))
    keep = np.nonzero(np.abs(lsf) > thr)[0]
    shifts = (n // 2 - h - keep) % n
    return shifts.astype(np.int64), lsf[keep].copy(), n


def convolve_1d_closed(line, lsf, rel_threshold=0.0):
    """Closed-form evaluation of convolve_1d (see ``lsf_taps``)."""
    line = np.asarray(line, dtype=np.float64)
    depth = line.shape[0]
    shifts, weights, n = lsf_taps(lsf, rel_threshold)
    ext = np.zeros(n)
    ext[:depth] = line
    k = np.arange(depth)
    out = np.zeros(depth)
    for s, wgt in zip(shifts, weights):
        out += wgt * ext[(k + s) % n]
    return out


def spectral_convolve(line, lsf):
    """lib/run.py:675-682: identity when lsf is None, else convolve_1d."""
    if lsf is None:
        return np.asarray(line, dtype=np.float64)
    return convolve_1d_fft(line, lsf)


# --------------------------------------------------------------------------- #
# a4 -- contribution of one spaxel                                             #
# --------------------------------------------------------------------------- #


def window_limits(y, x, H, W, fh, fw):
    """lib/run.py:407-410 and :697-706: cube window and matching FSF window."""
    fhh = (fh - 1) // 2
    fhw = (fw - 1) // 2
    y_min = max(y - fhh, 0)
    y_max = min(y + fhh + 1, H)
    x_min = max(x - fhw, 0)
    x_max = min(x + fhw + 1, W)
    ly0 = max(fhh - y, 0)
    ly1 = min(H + fhh - y, fh)
    lx0 = max(fhw - x, 0)
    lx1 = min(W + fhw - x, fw)
    return (y_min, y_max, x_min, x_max), (ly0, ly1, lx0, lx1)


def local_contribution(params, D, fsf, lsf):
    """lib/run.py:672-693: line -> LSF -> outer product with the FSF, [D,fh,fw]."""
    line = gaussian_line(np.arange(D), params[0], params[1], params[2])
    line_conv = spectral_convolve(line, lsf)
    return fsf * line_conv[:, np.newaxis][:, np.newaxis]


def contribution_of_spaxel(x, y, params, W, H, D, fsf, lsf):
    """lib/run.py:654-708 (reference-faithful: full zero cube + clipped paste)."""
    sim = np.zeros((D, H, W))
    local = local_contribution(params, D, fsf, lsf)
    (y0, y1, x0, x1), (ly0, ly1, lx0, lx1) = window_limits(
        y, x, H, W, fsf.shape[0], fsf.shape[1])
    sim[:, y0:y1, x0:x1] = local[:, ly0:ly1, lx0:lx1]
    return sim


def spaxel_iterator(mask):
    """lib/run.py:553-566 row-major over mask == 1."""
    H, W = mask.shape
    for y in range(H):
        for x in range(W):
            if mask[y, x] == 1:
                yield (y, x)


# --------------------------------------------------------------------------- #
# a11 / a12 -- full forward model                                              #
# --------------------------------------------------------------------------- #


def simulate_clean(shape, params, mask):
    """lib/run.py:597-621."""
    sim = np.zeros(shape)
    for (y, x) in spaxel_iterator(mask):
        sim[:, y, x] = gaussian_line(np.arange(shape[0]), *params[y, x])
    return sim


def lsf_lines(shape, params, mask, lsf):
    """lib/run.py:1011-1024: per-spaxel LSF-convolved lines, zero where masked."""
    sim = np.zeros(shape)
    for (y, x) in spaxel_iterator(mask):
        line = gaussian_line(np.arange(shape[0]), *params[y, x])
        sim[:, y, x] = spectral_convolve(line, lsf)
    return sim


def spatial_convolve(cube, fsf):
    """lib/run.py:1027-1029: scipy convolve2d(..., mode='same') per channel."""
    out = np.empty_like(cube)
    for z in range(cube.shape[0]):
        out[z] = convolve2d(cube[z], fsf, mode='same')
    return out


def convolve_cube(cube, fsf, lsf):
    """LSF along z of every spectrum, then FSF over (y,x) of every channel."""
    D, H, W = cube.shape
    tmp = np.empty_like(cube)
    for y in range(H):
        for x in range(W):
            tmp[:, y, x] = spectral_convolve(cube[:, y, x], lsf)
    return spatial_convolve(tmp, fsf)


# ---- the OLDER revision's convolution (lib/convolution.py:13-86, commented out
# in v0.3.0): a 3-D circular FFT convolution of the cube, padded to powers of two
# on every axis, with a 3-D PSF = LSF (x) FSF.  Restated because the reference's
# own saved cube pair (tests/input/GalPaK_*_myrun100k_{convolved,deconvolved}
# _cube.fits) was written by it: with a cube-sized Gaussian FSF image and
# sigma = fwhm / 2.35482 for BOTH spread functions it reproduces the pair to
# 1e-15 (tests/test_oracle.py), which pins padding rule, LSF vector, centring
# (fftshift + crop) and normalisation of the live 1-D code -- same pipeline,
# one axis -- against reference-written data.

def legacy_pad_cube(cube, axes=(0, 1, 2)):
    """lib/convolution.py:50-86 (pad_cube): zero padding to 2**len(bin(n-1)) with
    the data at offset diff/2 (+1 when diff is odd) -- the rule of `padding`,
    lib/convolution.py:137-155, on several axes."""
    old = cube.shape
    new = list(old)
    for ax in axes:
        new[ax] = padded_length(old[ax])
    slices = [slice(0, n) for n in old]
    for ax in axes:
        slices[ax] = slice(padding_offset(old[ax]), old[ax] + padding_offset(old[ax]))
    padded = np.zeros(new)
    padded[tuple(slices)] = cube
    return padded, tuple(slices)


def legacy_convolve_3d_same(cube, psf):
    """lib/convolution.py:13-47 (convolve_3d_same): rfftn * rfftn -> irfftn ->
    fftshift -> crop, on the padded grid (circular: "has edge effects")."""
    padded, slices = legacy_pad_cube(cube)
    size = padded.shape
    padded_psf, _ = legacy_pad_cube(psf)
    axes = (0, 1, 2)
    fft_psf = np.fft.rfftn(padded_psf, s=size, axes=axes)
    fft_img = np.fft.rfftn(padded, s=size, axes=axes)
    out = np.real(np.fft.fftshift(np.fft.irfftn(fft_img * fft_psf, s=size, axes=axes), axes=axes))
    return out[slices]


def legacy_convolve_2d_same(image, psf):
    """One channel of legacy_convolve_3d_same (the spatial factor of the
    separable 3-D PSF): padded circular 2-D FFT convolution."""
    return legacy_convolve_3d_same(image[None], psf[None])[0]


def legacy_gaussian_fsf_full(shape, fwhm_px):
    """The FSF image the older revision fed to convolve_3d_same: a Gaussian on the
    cube's full spatial grid (as MoffatFieldSpreadFunction still does,
    lib/spread_functions.py:165-189), centre (n-1)//2 - (n%2 - 1)
    (lib/spread_functions.py:107-110), sigma = fwhm / 2.35482 (the constant of
    lib/spread_functions.py:247), normalised to sum 1."""
    h, w = shape
    yo = (h - 1) // 2 - (h % 2 - 1)
    xo = (w - 1) // 2 - (w % 2 - 1)
    sigma = fwhm_px / 2.35482
    y, x = np.indices(shape)
    img = np.exp(-0.5 * ((y - yo) ** 2 + (x - xo) ** 2) / sigma ** 2)
    return img / img.sum()


def forward_full(shape, params, mask, fsf, lsf):
    """lib/run.py:999-1029 without the final ``data - sim``."""
    return spatial_convolve(lsf_lines(shape, params, mask, lsf), fsf)


def compute_error_in_one_step(data, params, mask, fsf, lsf):
    """lib/run.py:999-1031."""
    return data - forward_full(data.shape, params, mask, fsf, lsf)


def simulate_convolved(shape, params, mask, fsf, lsf):
    """lib/run.py:623-652: sum over spaxels of contribution_of_spaxel."""
    sim = np.zeros(shape)
    D, H, W = shape
    for (y, x) in spaxel_iterator(mask):
        sim = sim + contribution_of_spaxel(x, y, params[y][x], W, H, D, fsf, lsf)
    return sim


# --------------------------------------------------------------------------- #
# a6 / a9 -- windowed statistics                                               #
# --------------------------------------------------------------------------- #


def half_chi2(err_part, var_part):
    """lib/run.py:423-424  0.5 * nansum(err**2 / var)."""
    return 0.5 * np.nansum(err_part ** 2 / var_part)


def chi2_map(err, var):
    """Per-spaxel 0.5*nansum_z(err^2/var) -- the device chi2 map, [H,W]."""
    return 0.5 * np.nansum(err ** 2 / var, axis=0)


def gibbs_moments(ek_part, ul_part, var_part, ra):
    """lib/run.py:491-493: (ro, mu, sum ek^2/var, sum ek*ul/var)."""
    s_ee = np.sum(ek_part ** 2 / var_part)
    s_eu = np.sum(ek_part * ul_part / var_part)
    ro = ra / (1. + ra * s_ee)
    mu = ro * s_eu
    return ro, mu, s_ee, s_eu


def window_stats(err, var, params_yx, p_new, y, x, fsf, lsf):
    """
    The five numbers of the parity probe ``d3d_window_stats`` for a proposal
    ``p_new`` at spaxel (y, x), evaluated exactly like lib/run.py:391-426 and
    :464-493 would with ``contributions[y,x]`` rebuilt from the current
    parameters (memory-sane form):

        (ar_old, ar_new, delta, sum ek^2/var, sum ek*ul/var)

    ek is the unit-amplitude contribution of the *current* parameters.
    """
    D, H, W = err.shape
    fh, fw = fsf.shape
    (y0, y1, x0, x1), (ly0, ly1, lx0, lx1) = window_limits(y, x, H, W, fh, fw)
    c_old = local_contribution(params_yx, D, fsf, lsf)[:, ly0:ly1, lx0:lx1]
    c_new = local_contribution(p_new, D, fsf, lsf)[:, ly0:ly1, lx0:lx1]
    e_old = err[:, y0:y1, x0:x1]
    v = var[:, y0:y1, x0:x1]
    ul = e_old + c_old
    e_new = ul - c_new
    ar_old = half_chi2(e_old, v)
    ar_new = half_chi2(e_new, v)
    p_one = np.array(params_yx, dtype=np.float64).copy()
    p_one[0] = 1.
    ek = local_contribution(p_one, D, fsf, lsf)[:, ly0:ly1, lx0:lx1]
    _, _, s_ee, s_eu = gibbs_moments(ek, ul, v, 1.0)
    return np.array([ar_old, ar_new, ar_old - ar_new, s_ee, s_eu])


# --------------------------------------------------------------------------- #
# a13 -- spread-function taps (pixel units; unit handling is host-side)         #
# --------------------------------------------------------------------------- #


def _radius(xo, yo, x, y, pa, ba):
    """lib/spread_functions.py:113-131."""
    dx = xo - x
    dy = yo - y
    radian_pa = np.radians(pa)
    dx_p = dx * np.cos(radian_pa) - dy * np.sin(radian_pa)
    dy_p = dx * np.sin(radian_pa) + dy * np.cos(radian_pa)
    return np.sqrt(dx_p ** 2 + dy_p ** 2 / ba ** 2)


def _center(n):
    """lib/spread_functions.py:107-109 / :251  (n-1)//2 - (n%2 - 1)."""
    return (n - 1) // 2 - (n % 2 - 1)


def gaussian_fsf_image(fwhm_px, pa=0., ba=1.0):
    """lib/spread_functions.py:94-111 with the FWHM already in pixels."""
    stddev = fwhm_px / (2 * math.sqrt(2 * math.log(2)))
    size = int(math.ceil(6. * stddev))
    if size % 2 == 0:
        size += 1
    shape = (size, size)
    xo = _center(shape[1])
    yo = _center(shape[0])
    y, x = np.indices(shape)
    r = _radius(xo, yo, x, y, pa, ba)
    fsf = np.exp(-0.5 * (r / stddev) ** 2)
    return fsf / fsf.sum()


def moffat_fsf_image(shape, beta, fwhm_px=None, alpha_px=None, pa=0., ba=1.0):
    """lib/spread_functions.py:165-189 (image has the given spatial shape)."""
    xo = _center(shape[1])
    yo = _center(shape[0])
    y, x = np.indices(shape)
    r = _radius(xo, yo, x, y, pa, ba)
    if alpha_px is None:
        alpha = fwhm_px / (2. * np.sqrt(2. ** (1. / beta) - 1))
    else:
        alpha = alpha_px
    psf = (1. + (r / alpha) ** 2) ** (-beta)
    return psf / psf.sum()


def gaussian_lsf_vector(depth, sigma_px):
    """lib/spread_functions.py:245-261 with sigma already in pixels."""
    zc = _center(depth)
    z_range = np.arange(depth) - zc
    if sigma_px == 0:
        lsf = np.zeros(depth)
        lsf[zc] = 1.
    else:
        lsf = np.exp((z_range - 0.) ** 2 / (-2. * sigma_px ** 2))
    return lsf / lsf.sum()


# --------------------------------------------------------------------------- #
# a14 -- default noise estimate                                                #
# --------------------------------------------------------------------------- #


def median_clip(data, clip_sigma=3., limit_ratio=1e-3, max_iterations=5):
    """lib/math_utils.py:16-57."""
    data = data[(np.isnan(data) == False) * np.isfinite(data)]  # noqa: E712
    median = np.median(data)
    iteration = 0
    finished = False
    while not finished:
        iteration += 1
        lastct = median
        median = np.median(data)
        sigma = np.std(data)
        index = np.nonzero(np.abs(data - median) < clip_sigma * sigma)
        if np.size(index) > 0:
            data = data[index]
        if (abs(median - lastct) / abs(lastct) < limit_ratio) \
                or (iteration >= max_iterations):
            finished = True
    median = np.median(data)
    sigma = np.std(data)
    return median, sigma, iteration


def default_variance(data):
    """lib/run.py:186-192."""
    sub = np.copy(data[2:-2, 2:-4, 2:4])
    _, clip_sigma, _ = median_clip(sub, 2.5)
    if clip_sigma == 0:
        clip_sigma = 1e-20
    return np.ones(data.shape) * clip_sigma ** 2


# --------------------------------------------------------------------------- #
# Random numbers shared with the device (NOT in the reference, which uses the  #
# unseeded global numpy RNG: lib/run.py:313,435,578; lib/rtnorm.py:17).        #
# The device draws from Philox4x32-10 keyed by (seed, spaxel, sweep, block);   #
# the oracle restates that generator so chains can be compared update by       #
# update.                                                                      #
# --------------------------------------------------------------------------- #

_M0 = 0xD2511F53
_M1 = 0xCD9E8D57
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK32 = 0xFFFFFFFF


def philox4x32_10(counter, key):
    """Philox4x32-10 (Salmon et al., SC'11).  counter: 4 u32, key: 2 u32."""
    c0, c1, c2, c3 = [int(v) & _MASK32 for v in counter]
    k0, k1 = [int(v) & _MASK32 for v in key]
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> 32, p0 & _MASK32
        hi1, lo1 = p1 >> 32, p1 & _MASK32
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & _MASK32, lo1, (hi0 ^ c3 ^ k1) & _MASK32, lo0
        k0 = (k0 + _W0) & _MASK32
        k1 = (k1 + _W1) & _MASK32
    return c0, c1, c2, c3


def u64_to_unit(u):
    """(0,1) double from 64 random bits: ((u >> 12) + 0.5) * 2**-52 (exact in
    fp64: the result lies in [2**-53, 1 - 2**-53], never 0 or 1)."""
    return ((u >> 12) + 0.5) * (1.0 / 4503599627370496.0)


def philox_pair(seed, spaxel, sweep, block):
    """Two (0,1) doubles for (seed, global spaxel index, sweep, block)."""
    r = philox4x32_10((spaxel, sweep, block, 0),
                      (seed & _MASK32, (seed >> 32) & _MASK32))
    return (u64_to_unit((r[1] << 32) | r[0]), u64_to_unit((r[3] << 32) | r[2]))


TN_TAIL = 6.0  # beyond this many sigmas use exponential rejection
_SQRT2 = math.sqrt(2.0)


def truncated_standard_normal(alpha, beta, draw):
    """
    One draw of N(0,1) truncated to [alpha, beta] (alpha < beta).
    ``draw()`` returns the next pair of (0,1) uniforms.  Own sampler (the
    reference's is Chopin's table method, lib/rtnorm.py:95-224, GPL tables not
    reproduced): inverse CDF on the side of the interval where it is well
    conditioned, Robert's (1995) translated-exponential rejection beyond
    TN_TAIL sigmas.  Same distribution as ``rtstdnorm``; checked by KS tests
    against draws of the reference's ``rtnorm``.
    """
    if alpha > beta:
        raise ValueError("alpha must be < beta")
    if beta <= 0.0:
        return -truncated_standard_normal(-beta, -alpha, draw)
    if alpha >= TN_TAIL:
        lam = 0.5 * (alpha + math.sqrt(alpha * alpha + 4.0))
        for _ in range(1000):
            u1, u2 = draw()
            z = alpha - math.log(u1) / lam
            if z <= beta and math.log(u2) <= -0.5 * (z - lam) ** 2:
                return z
        return alpha
    u1, _ = draw()
    if alpha > 0.0:
        qa = 0.5 * erfc(alpha / _SQRT2)
        qb = 0.5 * erfc(beta / _SQRT2)
        q = qa - u1 * (qa - qb)
        z = _SQRT2 * erfcinv(2.0 * q)
    else:
        pa = ndtr(alpha)
        pb = ndtr(beta)
        z = ndtri(pa + u1 * (pb - pa))
    return float(min(max(z, alpha), beta))


def truncated_normal(lo, hi, mu, sigma, draw):
    """TN(lo, hi; mu, sigma) -- distribution of lib/rtnorm.py:21-92 ``rtnorm``."""
    alpha = (lo - mu) / sigma
    beta = (hi - mu) / sigma
    z = truncated_standard_normal(alpha, beta, draw)
    return float(min(max(mu + sigma * z, lo), hi))


# --------------------------------------------------------------------------- #
# a5-a10 -- one MH-within-Gibbs spaxel update, memory-sane form                #
# --------------------------------------------------------------------------- #

# Philox block layout of one spaxel update (shared with the device kernel):
BLK_JUMP_AC = 0   # (u_a, u_c)
BLK_JUMP_W = 1    # (u_w, u_accept)
BLK_GIBBS = 2     # first truncated-normal block; rejection trials use 3, 4, ...


class MHState(object):
    """Mutable chain state of the oracle sampler."""

    def __init__(self, data, var, mask, fsf, lsf, params, min_b, max_b,
                 jump_amplitude=0.1, gibbs_apriori_variance=None, seed=12345,
                 origin=None, err=None):
        self.data = np.asarray(data, dtype=np.float64)
        self.var = np.asarray(var, dtype=np.float64)
        self.mask = np.asarray(mask)
        self.fsf = np.asarray(fsf, dtype=np.float64)
        self.lsf = None if lsf is None else np.asarray(lsf, dtype=np.float64)
        self.params = np.array(params, dtype=np.float64)
        self.min_b = np.asarray(min_b, dtype=np.float64)
        self.max_b = np.asarray(max_b, dtype=np.float64)
        # (gy0, gx0, Wg): this state is a tile of a wider cube; Philox keys use
        # the GLOBAL spaxel index (multi-GPU tiling).  None: the whole cube.
        self.origin = origin if origin is not None else (0, 0, self.data.shape[2])
        self.last = None  # (p_old, p_end) of the last update
        amp = np.ones(3) * np.array(jump_amplitude)   # lib/run.py:251-252
        amp[0] = 0.                                   # lib/run.py:262
        self.amp = amp
        if gibbs_apriori_variance is None:            # lib/run.py:264-265
            gibbs_apriori_variance = float(self.max_b[0] ** 2)
        self.ra = gibbs_apriori_variance
        self.seed = seed
        self.err = np.array(err, dtype=np.float64) if err is not None else \
            compute_error_in_one_step(self.data, self.params, self.mask, self.fsf, self.lsf)
        self.accepted = 0
        self.dlog = np.zeros(self.mask.shape)


def colour_order(mask, fh, fw):
    """
    The device scan order: colours (y mod fh, x mod fw) in row-major colour
    order, spaxels row-major inside a colour.  Same-colour windows are
    disjoint, so this sequential order equals the parallel device update.
    The reference declares the scan order overridable (lib/run.py:553-560).
    """
    H, W = mask.shape
    for cy in range(fh):
        for cx in range(fw):
            for y in range(cy, H, fh):
                for x in range(cx, W, fw):
                    if mask[y, x] == 1:
                        yield (y, x)


def mh_update(st, y, x, sweep):
    """
    One spaxel update: lib/run.py:369-519 with the ``contributions`` array
    replaced by a rebuild from the current parameters (memory-sane form) and
    the numpy global RNG replaced by the Philox stream.  Returns True when the
    MH proposal was accepted.
    """
    D, H, W = st.data.shape
    fh, fw = st.fsf.shape
    gy0, gx0, Wg = st.origin
    sp = (y + gy0) * Wg + (x + gx0)      # global spaxel index keys the RNG
    p_old = st.params[y, x].copy()

    # lib/run.py:570-579 Cauchy jump
    ua, uc = philox_pair(st.seed, sp, sweep, BLK_JUMP_AC)
    uw, uacc = philox_pair(st.seed, sp, sweep, BLK_JUMP_W)
    u = np.array([ua, uc, uw])
    p_new = p_old + st.amp * np.tan(np.pi * (u - 0.5))

    # lib/run.py:379-388 (gibbs on: OOB proposals are evaluated then rejected)
    oob = bool((p_new < st.min_b).any() or (p_new > st.max_b).any())

    (y0, y1, x0, x1), (ly0, ly1, lx0, lx1) = window_limits(y, x, H, W, fh, fw)
    c_new = local_contribution(p_new, D, st.fsf, st.lsf)[:, ly0:ly1, lx0:lx1]
    c_old = local_contribution(p_old, D, st.fsf, st.lsf)[:, ly0:ly1, lx0:lx1]
    e_old = st.err[:, y0:y1, x0:x1]
    v = st.var[:, y0:y1, x0:x1]

    ul = e_old + c_old                       # lib/run.py:400
    e_new = ul - c_new                       # lib/run.py:402
    ar_old = half_chi2(e_old, v)             # lib/run.py:423
    ar_new = half_chi2(e_new, v)             # lib/run.py:424
    delta = ar_old - ar_new                  # lib/run.py:426
    st.dlog[y, x] = delta

    accepted = (math.log(uacc) < delta) and not oob   # lib/run.py:435-438
    p_end = p_new.copy() if accepted else p_old.copy()
    if accepted:
        st.accepted += 1

    # lib/run.py:456-519 Gibbs draw of the amplitude
    p_one = p_end.copy()
    p_one[0] = 1.
    ek = local_contribution(p_one, D, st.fsf, st.lsf)[:, ly0:ly1, lx0:lx1]
    ro, mu, _, _ = gibbs_moments(ek, ul, v, st.ra)
    blk = [BLK_GIBBS]

    def draw():
        pair = philox_pair(st.seed, sp, sweep, blk[0])
        blk[0] += 1
        return pair

    r = truncated_normal(st.min_b[0], st.max_b[0], mu, math.sqrt(ro), draw)
    p_end[0] = r
    st.err[:, y0:y1, x0:x1] = ul - ek * r    # lib/run.py:508-515
    st.params[y, x] = p_end
    st.last = (p_old, p_end.copy())
    return accepted


def replay_update(st, y, x, p_old, p_end):
    """
    Residual change of an update made by ANOTHER tile at local (y, x) (possibly
    outside this tile): the same (e + c_old) - ek*r of mh_update on the part of
    the window inside this tile.  Not in the reference (single process).
    """
    D, H, W = st.data.shape
    fh, fw = st.fsf.shape
    fhh, fhw = (fh - 1) // 2, (fw - 1) // 2
    y0, y1 = max(y - fhh, 0), min(y + fhh + 1, H)
    x0, x1 = max(x - fhw, 0), min(x + fhw + 1, W)
    if y0 >= y1 or x0 >= x1:
        return
    ly0, ly1 = y0 - (y - fhh), y1 - (y - fhh)
    lx0, lx1 = x0 - (x - fhw), x1 - (x - fhw)
    c_old = local_contribution(p_old, D, st.fsf, st.lsf)[:, ly0:ly1, lx0:lx1]
    p_one = np.array(p_end, dtype=np.float64).copy()
    p_one[0] = 1.
    ek = local_contribution(p_one, D, st.fsf, st.lsf)[:, ly0:ly1, lx0:lx1]
    st.err[:, y0:y1, x0:x1] = (st.err[:, y0:y1, x0:x1] + c_old) - ek * p_end[0]
    if 0 <= y < H and 0 <= x < W:
        st.params[y, x] = p_end


def mh_sweep(st, sweep, order=None):
    """One sweep in device colour order (or a given order)."""
    fh, fw = st.fsf.shape
    if order is None:
        order = colour_order(st.mask, fh, fw)
    n = 0
    for (y, x) in order:
        mh_update(st, y, x, sweep)
        n += 1
    return n


# --------------------------------------------------------------------------- #
# Reference-faithful sweep (B-ref): full-cube temporaries + contributions      #
# array, lib/run.py:285-288, 317-334, 367-519.  Only feasible for small cubes. #
# --------------------------------------------------------------------------- #


class RefFaithfulState(object):
    def __init__(self, data, var, mask, fsf, lsf, params, min_b, max_b,
                 jump_amplitude=0.1, gibbs_apriori_variance=None, seed=12345):
        self.data = np.asarray(data, dtype=np.float64)
        self.var = np.asarray(var, dtype=np.float64)
        self.mask = np.asarray(mask)
        self.fsf = np.asarray(fsf, dtype=np.float64)
        self.lsf = lsf
        self.params = np.array(params, dtype=np.float64)
        self.min_b = np.asarray(min_b, dtype=np.float64)
        self.max_b = np.asarray(max_b, dtype=np.float64)
        amp = np.ones(3) * np.array(jump_amplitude)
        amp[0] = 0.
        self.amp = amp
        if gibbs_apriori_variance is None:
            gibbs_apriori_variance = float(self.max_b[0] ** 2)
        self.ra = gibbs_apriori_variance
        self.seed = seed
        D, H, W = self.data.shape
        self.contributions = np.zeros((H, W, D, H, W))   # lib/run.py:285-288
        sim = np.zeros_like(self.data)
        for (y, x) in spaxel_iterator(self.mask):        # lib/run.py:322-331
            c = contribution_of_spaxel(x, y, self.params[y][x], W, H, D,
                                       self.fsf, self.lsf)
            sim = sim + c
            self.contributions[y, x, :, :, :] = c
        self.err = self.data - sim                       # lib/run.py:334
        self.accepted = 0
        self.dlog = np.zeros(self.mask.shape)


def ref_faithful_update(st, y, x, sweep):
    """lib/run.py:369-519, statement for statement (Philox instead of np.random)."""
    D, H, W = st.data.shape
    fh, fw = st.fsf.shape
    fhh = (fh - 1) // 2
    fhw = (fw - 1) // 2
    sp = y * W + x
    p_old = np.array(st.params[y][x].tolist())
    ua, uc = philox_pair(st.seed, sp, sweep, BLK_JUMP_AC)
    uw, uacc = philox_pair(st.seed, sp, sweep, BLK_JUMP_W)
    p_new = p_old + st.amp * np.tan(np.pi * (np.array([ua, uc, uw]) - 0.5))
    out_of_bounds = bool((p_new < st.min_b).any() or (p_new > st.max_b).any())
    contribution = contribution_of_spaxel(x, y, p_new, W, H, D, st.fsf, st.lsf)
    ul = st.err + st.contributions[y, x]
    err_new = ul - contribution
    y_min = max(y - fhh, 0)
    y_max = min(y + fhh + 1, H)
    x_min = max(x - fhw, 0)
    x_max = min(x + fhw + 1, W)
    err_new_part = err_new[:, y_min:y_max, x_min:x_max]
    err_old_part = st.err[:, y_min:y_max, x_min:x_max]
    var_part = st.var[:, y_min:y_max, x_min:x_max]
    ar_part_old = 0.5 * np.nansum(err_old_part ** 2 / var_part)
    ar_part_new = 0.5 * np.nansum(err_new_part ** 2 / var_part)
    cur_acceptance = ar_part_old - ar_part_new
    st.dlog[y, x] = cur_acceptance
    min_acceptance = math.log(uacc)
    if min_acceptance < cur_acceptance and not out_of_bounds:
        st.contributions[y, x, :, :, :] = contribution
        st.accepted += 1
        st.err = err_new
        p_end = p_new.copy()
        accepted = True
    else:
        p_end = p_old.copy()
        accepted = False
    st.params[y][x] = p_end
    gibbsed_value = st.params[y][x][0]
    ul_part = ul[:, y_min:y_max, x_min:x_max]
    ek_part = st.contributions[y, x, :, y_min:y_max, x_min:x_max]
    if gibbsed_value != 0:
        ek_part = ek_part / gibbsed_value
    else:
        p_one = st.params[y][x].copy()
        p_one[0] = 1.
        contribution_one = contribution_of_spaxel(x, y, p_one, W, H, D,
                                                  st.fsf, st.lsf)
        ek_part = contribution_one[:, y_min:y_max, x_min:x_max]
    ra = st.ra
    ro = ra / (1. + ra * np.sum(ek_part ** 2 / var_part))
    mu = ro * np.sum(ek_part * ul_part / var_part)
    blk = [BLK_GIBBS]

    def draw():
        pair = philox_pair(st.seed, sp, sweep, blk[0])
        blk[0] += 1
        return pair

    r = truncated_normal(st.min_b[0], st.max_b[0], mu, np.sqrt(ro), draw)
    p_end[0] = r
    contribution = np.zeros(st.data.shape)
    contribution[:, y_min:y_max, x_min:x_max] = ek_part * r
    err_new = ul - contribution
    st.contributions[y, x, :, :, :] = contribution
    st.err = err_new
    st.params[y][x] = p_end
    return accepted


# --------------------------------------------------------------------------- #
# Synthetic inputs of SURVEY.md 8(d)                                           #
# --------------------------------------------------------------------------- #


def moffat_cropped(size, fwhm_px, beta):
    """
    'Moffat PSF radius 5' of BASELINE config 2/3: the reference has no
    truncation parameter, so the image is evaluated on a size x size grid with
    the reference formula (lib/spread_functions.py:165-189) and used through
    ImageFieldSpreadFunction semantics (:55-65).  Normalised to sum 1.
    """
    return moffat_fsf_image((size, size), beta, fwhm_px=fwhm_px)


def muse_like_lsf(depth, sigma_px=0.9, box_px=1.0):
    """
    Analytic stand-in for MUSELineSpreadFunction (mpdaf absent; parity
    unpinned): pixel-integrated box (slit) convolved with a Gaussian, centred
    like lib/spread_functions.py:251, normalised to sum 1.
    """
    zc = _center(depth)
    z = np.arange(depth) - zc
    s = sigma_px * _SQRT2
    h = box_px / 2.0
    from scipy.special import erf
    prof = 0.5 * (erf((z + h) / s) - erf((z - h) / s))
    prof[np.abs(prof) < 1e-300] = 0.0
    return prof / prof.sum()


def synthetic_truth(D, H, W, rng, A0=10.0):
    """Truth parameter map of SURVEY.md 8(d)."""
    y, x = np.indices((H, W))
    r2 = (y - H / 2.) ** 2 + (x - W / 2.) ** 2
    a = A0 * np.exp(-r2 / (2. * (H / 6.) ** 2))
    c = D / 2. + (D / 8.) * np.tanh((x - W / 2.) / (W / 8.))
    w = rng.uniform(1.5, 3.0, size=(H, W))
    return np.dstack((a, c, w))


def synthetic_case(D, H, W, fsf, lsf, seed=12345, A0=10.0, fast_forward=None):
    """
    (data, var, mask, truth, initial params, min_b, max_b) of SURVEY.md 8(d).
    ``fast_forward`` may supply a forward operator for big cubes (the GPU one
    in bench.py); default is the oracle's.
    """
    rng = np.random.default_rng(seed)
    truth = synthetic_truth(D, H, W, rng, A0)
    mask = np.ones((H, W))
    if fast_forward is None:
        clean = forward_full((D, H, W), truth, mask, fsf, lsf)
    else:
        clean = fast_forward(truth)
    sigma = 0.05 * A0 * np.max(fsf)
    data = clean + rng.normal(0., sigma, size=(D, H, W))
    var = np.ones((D, H, W)) * sigma ** 2
    min_b = model_min_boundaries()
    max_b = model_max_boundaries(data, fsf)
    init = min_b + (max_b - min_b) * rng.random((H, W, 3))   # lib/run.py:310-314
    return data, var, mask, truth, init, min_b, max_b
