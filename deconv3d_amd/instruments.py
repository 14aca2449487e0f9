# coding=utf-8
"""Instrument containers -- reference: lib/instruments.py."""
from .cube import Axis, Cube
from .spread_functions import (FieldSpreadFunction, GaussianFieldSpreadFunction,
                               GaussianLineSpreadFunction, LineSpreadFunction)


class Instrument:
    """Holds the LSF and FSF plugins (lib/instruments.py:11-34)."""

    def __init__(self, lsf, fsf):
        if not isinstance(lsf, LineSpreadFunction):
            raise ValueError("lsf= MUST be an instance of LineSpreadFunction")
        self.lsf = lsf
        if not isinstance(fsf, FieldSpreadFunction):
            raise ValueError("fsf= MUST be an instance of FieldSpreadFunction")
        self.fsf = fsf

    def __str__(self):
        return "\nfsf = %s\nlsf = %s\n" % (self.fsf, self.lsf)


class MUSE(Instrument):
    """MUSE defaults: Gaussian LSF FWHM 2.675 A, Gaussian FSF FWHM 1"
    (lib/instruments.py:103-119)."""

    def __init__(self, lsf=None, fsf=None, lsf_fwhm=0.0002675,
                 fsf_fwhm=1.0, fsf_pa=0., fsf_ba=1.0):
        if lsf is None:
            lsf = GaussianLineSpreadFunction(fwhm=lsf_fwhm)
        if fsf is None:
            fsf = GaussianFieldSpreadFunction(fwhm=fsf_fwhm, pa=fsf_pa, ba=fsf_ba)
        Instrument.__init__(self, lsf=lsf, fsf=fsf)

    def build_cube(self, data):
        """Wrap a bare ndarray with MUSE WCS metadata: 0.2" spaxels, 1.25 A
        channels (lib/instruments.py:121-151)."""
        meta = {
            'CDELT1': 5.5555555555555e-05, 'CDELT2': 5.5555555555555e-05, 'CDELT3': 1.25,
            'CRVAL1': 1.0, 'CRVAL2': 1.0, 'CRVAL3': 6564.0,
            'CRPIX1': 1.0, 'CRPIX2': 1.0, 'CRPIX3': 15.0,
            'CUNIT1': 'deg', 'CUNIT2': 'deg', 'CUNIT3': 'Angstrom',
            'CTYPE1': 'RA---TAN', 'CTYPE2': 'DEC--TAN',
        }
        x = Axis('x', meta['CRVAL1'], meta['CDELT1'], meta['CUNIT1'])
        y = Axis('y', meta['CRVAL2'], meta['CDELT2'], meta['CUNIT2'])
        z = Axis('z', meta['CRVAL3'], meta['CDELT3'], meta['CUNIT3'])
        z.crpix = meta['CRPIX3']
        return Cube(data=data, meta={'fits': meta}, x=x, y=y, z=z)
