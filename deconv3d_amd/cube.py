# coding=utf-8
"""
Minimal hyperspectral cube container.

The reference uses the third-party ``hyperspectral.HyperspectralCube`` (PyPI,
unpinned, absent here) only for I/O and metadata: ``.data``, ``.shape``,
``.is_empty()``, ``.meta``, ``.get_step(axis)`` (lib/spread_functions.py:96,247),
``Cube.from_fits`` (lib/run.py:121), ``.to_fits`` (lib/run.py:778-779) and
``Axis`` (lib/instruments.py:142-151).  This module provides exactly that
surface, with a tiny FITS reader/writer (single primary HDU, BITPIX -64/-32/
16/32, which is what the reference's fixtures are) and a tiny unit helper in
place of ``astropy.units``.
"""
from __future__ import annotations

import math

import numpy as np

# --------------------------------------------------------------------------- #
# units                                                                        #
# --------------------------------------------------------------------------- #

_LENGTH = {  # in metres
    "m": 1.0, "meter": 1.0, "metre": 1.0,
    "mm": 1e-3, "um": 1e-6, "micron": 1e-6, "microns": 1e-6, "µm": 1e-6,
    "nm": 1e-9, "angstrom": 1e-10, "angstroms": 1e-10, "aa": 1e-10, "a": 1e-10,
}
_ANGLE = {  # in degrees
    "deg": 1.0, "degree": 1.0, "degrees": 1.0,
    "arcmin": 1.0 / 60.0, "arcsec": 1.0 / 3600.0, "mas": 1.0 / 3.6e6,
    "rad": 180.0 / math.pi, "radian": 180.0 / math.pi,
}


def _unit_key(unit):
    return str(unit).strip().lower()


class Quantity(object):
    """``value`` with a unit string; ``.to(unit).value`` converts."""

    def __init__(self, value, unit):
        self.value = float(value)
        self.unit = str(unit).strip()

    def to(self, unit):
        src, dst = _unit_key(self.unit), _unit_key(unit)
        for table in (_LENGTH, _ANGLE):
            if src in table and dst in table:
                return Quantity(self.value * table[src] / table[dst], unit)
        raise ValueError("cannot convert unit '%s' to '%s'" % (self.unit, unit))

    def __repr__(self):
        return "%r %s" % (self.value, self.unit)


class Axis(object):
    """lib/instruments.py:142-144 ``Axis(name, start, step, unit)``."""

    def __init__(self, name, start, step, unit):
        self.name = name
        self.start = float(start)
        self.step = float(step)
        self.unit = str(unit).strip()


# --------------------------------------------------------------------------- #
# FITS (primary HDU only)                                                      #
# --------------------------------------------------------------------------- #

_BITPIX = {-64: ">f8", -32: ">f4", 8: "u1", 16: ">i2", 32: ">i4", 64: ">i8"}


def _parse_card_value(raw):
    raw = raw.split("/")[0].strip() if not raw.strip().startswith("'") else raw.strip()
    if raw.startswith("'"):
        # a quote inside a FITS string is doubled; the string ends at the first single one
        out, i = [], 1
        while i < len(raw):
            if raw[i] == "'":
                if i + 1 < len(raw) and raw[i + 1] == "'":
                    out.append("'")
                    i += 2
                    continue
                break
            out.append(raw[i])
            i += 1
        return "".join(out).rstrip()
    if raw in ("T", "F"):
        return raw == "T"
    try:
        return int(raw)
    except ValueError:
        try:
            return float(raw.replace("D", "E"))
        except ValueError:
            return raw


def read_fits(path):
    """Returns (data ndarray in C order (NAXISn..NAXIS1), header dict)."""
    with open(path, "rb") as fh:
        raw = fh.read()
    header = {}
    pos = 0
    done = False
    while not done:
        block = raw[pos:pos + 2880]
        if len(block) < 2880:
            raise ValueError("truncated FITS header in %s" % path)
        pos += 2880
        for i in range(0, 2880, 80):
            card = block[i:i + 80].decode("latin1")
            key = card[:8].strip()
            if key == "END":
                done = True
                break
            if card[8:10] == "= ":
                header[key] = _parse_card_value(card[10:])
    naxis = int(header.get("NAXIS", 0))
    if naxis == 0:
        return None, header
    shape = tuple(int(header["NAXIS%d" % (k + 1)]) for k in range(naxis))[::-1]
    dtype = np.dtype(_BITPIX[int(header["BITPIX"])])
    count = int(np.prod(shape))
    data = np.frombuffer(raw, dtype=dtype, count=count, offset=pos).reshape(shape)
    data = data.astype(np.float64)
    if "BSCALE" in header or "BZERO" in header:
        data = data * float(header.get("BSCALE", 1.0)) + float(header.get("BZERO", 0.0))
    return data, header


def _card(key, value):
    """One 80-character header card, or None for a value FITS cannot hold
    (NaN / infinity: the standard has no representation for them)."""
    if isinstance(value, (float, np.floating)) and not np.isfinite(value):
        return None
    if isinstance(value, bool):
        v = "T" if value else "F"
        body = "%-8s= %20s" % (key, v)
    elif isinstance(value, (int, np.integer)):
        body = "%-8s= %20d" % (key, value)
    elif isinstance(value, (float, np.floating)):
        body = "%-8s= %20s" % (key, repr(float(value)).upper().replace("E+", "E"))
    else:
        # quotes doubled, text cut so that the closing quote stays inside the card
        text = str(value).replace("'", "''")[:68]
        if text.endswith("'") and (len(text) - len(text.rstrip("'"))) % 2:
            text = text[:-1]          # do not split a doubled quote
        body = "%-8s= %-20s" % (key, "'%-8s'" % text)
    return body[:80].ljust(80)


def write_fits(path, data, header=None, clobber=False):
    import os
    if os.path.exists(path) and not clobber:
        raise IOError("File '%s' exists (use clobber=True)" % path)
    data = np.asarray(data, dtype=np.float64)
    cards = [_card("SIMPLE", True), _card("BITPIX", -64), _card("NAXIS", data.ndim)]
    for k, n in enumerate(data.shape[::-1]):
        cards.append(_card("NAXIS%d" % (k + 1), int(n)))
    skip = {"SIMPLE", "BITPIX", "NAXIS", "EXTEND", "BSCALE", "BZERO"}
    for key, value in (header or {}).items():
        if key in skip or key.startswith("NAXIS"):
            continue
        card = _card(key, value)
        if card is not None:
            cards.append(card)
    cards.append("END".ljust(80))
    head = "".join(cards)
    head += " " * ((-len(head)) % 2880)
    body = data.astype(">f8").tobytes()
    body += b"\0" * ((-len(body)) % 2880)
    with open(path, "wb") as fh:
        fh.write(head.encode("latin1"))
        fh.write(body)


# --------------------------------------------------------------------------- #
# Cube                                                                         #
# --------------------------------------------------------------------------- #


class Cube(object):
    """
    Stand-in for ``hyperspectral.HyperspectralCube``: ``data`` is ``(D, H, W)``
    = (spectral z, y, x), x fastest (lib/run.py:146-149).
    """

    def __init__(self, data=None, meta=None, x=None, y=None, z=None):
        self.data = None if data is None else np.asarray(data, dtype=np.float64)
        self.meta = {} if meta is None else meta
        self.x, self.y, self.z = x, y, z

    @property
    def shape(self):
        return None if self.data is None else self.data.shape

    def is_empty(self):
        return self.data is None or self.data.size == 0

    @classmethod
    def from_fits(cls, path):
        data, header = read_fits(path)
        axes = {}
        for k, name in ((1, "x"), (2, "y"), (3, "z")):
            step = header.get("CDELT%d" % k, header.get("CD%d_%d" % (k, k)))
            if step is None:
                continue
            unit = header.get("CUNIT%d" % k, "deg" if k < 3 else "Angstrom")
            axes[name] = Axis(name, header.get("CRVAL%d" % k, 0.0), step, unit)
            axes[name].crpix = header.get("CRPIX%d" % k, 1.0)
        return cls(data=data, meta={"fits": header}, **axes)

    def to_fits(self, path, clobber=False):
        header = dict(self.meta.get("fits", {})) if isinstance(self.meta, dict) else {}
        for k, ax in ((1, self.x), (2, self.y), (3, self.z)):
            if ax is not None:
                header.setdefault("CDELT%d" % k, ax.step)
                header.setdefault("CUNIT%d" % k, ax.unit)
                header.setdefault("CRVAL%d" % k, ax.start)
        write_fits(path, self.data, header, clobber)

    def _axis(self, index):
        return (self.z, self.y, self.x)[index]

    def get_step(self, axis):
        """Step of axis 0 (z), 1 (y) or 2 (x) as a :class:`Quantity`."""
        ax = self._axis(axis)
        if ax is None:
            raise ValueError(
                "cube has no metadata for axis %d; build it with MUSE.build_cube(data) or "
                "Cube(data, x=Axis(...), y=Axis(...), z=Axis(...))" % axis)
        return Quantity(abs(ax.step), ax.unit)

    # used by MUSELineSpreadFunction (lib/spread_functions.py:303-306), in microns
    @property
    def z_step(self):
        return self.get_step(0).to("um").value

    @property
    def z_central(self):
        ax = self.z
        crpix = getattr(ax, "crpix", 1.0)
        mid = (self.data.shape[0] - 1) / 2.0
        return Quantity(ax.start + (mid - (crpix - 1.0)) * ax.step, ax.unit).to("um").value

    def __str__(self):
        return "Cube%s" % (self.shape,)


HyperspectralCube = Cube
