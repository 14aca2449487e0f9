# coding=utf-8
"""
MH-within-Gibbs chain for a line model evaluated on the HOST.

The device kernels evaluate ``SingleGaussianLineModel`` themselves.  Any other
``LineModel`` plugin (own ``modelize`` / ``post_jump`` / parameter count,
lib/line_models.py:17-61) goes through this slower path: per colour class the
host draws the Cauchy proposals (lib/run.py:570-579), calls the model's hooks
and ``modelize`` for the current and proposed parameters, and hands the UNIT
lines to ``d3d_mh_colour_lines``; the LSF, the FSF-window statistics, the
accept test, the Gibbs draw of the amplitude and the residual update stay on the
GPU (the same kernel code as the built-in model).  As in the reference the Gibbs
parameter is assumed to be a pure amplitude: line(p) = p[g] * line(p with p[g]=1)
(lib/run.py:458-488).
"""
from __future__ import annotations

import math

import numpy as np

from . import _lib


class HostModelChain(object):
    def __init__(self, run, engine, params, min_b, max_b, jump_amp, ra, seed, refresh_every):
        self.run, self.eng = run, engine
        self.model = run.model
        self.g = self.model.gibbs_parameter_index()
        self.D, self.H, self.W = run.cube.data.shape
        self.x = np.arange(self.D, dtype=np.float64)
        self.params = np.array(params, dtype=np.float64)
        self.min_b, self.max_b = np.asarray(min_b, float), np.asarray(max_b, float)
        self.amp = np.asarray(jump_amp, float)
        self.mask = run.mask
        self.refresh_every = int(refresh_every)
        self.seed = int(seed)
        self.sweep_origin = 0
        self.rng = np.random.Generator(np.random.Philox(int(seed)))
        fh, fw = run.fsf.shape
        self.colours = []
        for cy in range(fh):
            for cx in range(fw):
                ys, xs = np.nonzero(self.mask[cy::fh, cx::fw] == 1)
                self.colours.append((cy + ys * fh, cx + xs * fw))
        gb_lo = self.min_b[self.g] if self.g is not None else 0.0
        gb_hi = self.max_b[self.g] if self.g is not None else 1.0
        engine.mh_config([gb_lo, 0., 0.], [gb_hi, 1., 1.], [0., 0., 0.],
                         ra if self.g is not None else 1.0, seed=seed, refresh_every=0)
        self.data0 = np.where(np.isnan(run.cube.data), 0.0, run.cube.data)
        self.unit = np.zeros((self.H, self.W, self.D))
        for y in range(self.H):
            for x in range(self.W):
                if self.mask[y, x] == 1:
                    self.unit[y, x] = self.unit_line(self.params[y, x])
        self.refresh()

    def set_sweep_origin(self, origin):
        """Resumed run: the device draws of sweep s use the streams of sweep
        s + origin and the host proposals a generator keyed by (seed, origin)."""
        self.sweep_origin = int(origin)
        self.rng = np.random.Generator(np.random.Philox(key=[self.seed & (2 ** 64 - 1),
                                                             self.sweep_origin]))

    # -- model evaluation ---------------------------------------------------
    def unit_line(self, p):
        q = np.array(p, dtype=np.float64)
        if self.g is not None:
            q[self.g] = 1.0
        return np.asarray(self.model.modelize(self.run, self.x, q), dtype=np.float64)

    def clean_cube(self, params=None):
        """Cube of the raw lines of the unmasked spaxels (lib/run.py:597-621)."""
        cube = np.zeros((self.D, self.H, self.W))
        for y in range(self.H):
            for x in range(self.W):
                if self.mask[y, x] != 1:
                    continue
                if params is None:
                    amp = self.params[y, x, self.g] if self.g is not None else 1.0
                    cube[:, y, x] = amp * self.unit[y, x]
                else:
                    cube[:, y, x] = np.asarray(
                        self.model.modelize(self.run, self.x, params[y, x]), dtype=np.float64)
        return cube

    def refresh(self):
        """err = data - LSF (x) FSF (model), lib/run.py:334 and :521-534."""
        sim = self.eng.convolve(self.clean_cube())
        self.eng.upload_slot(_lib.SLOT_ERR, self.data0 - sim)

    # -- one sweep ----------------------------------------------------------
    def sweep(self, s, dlog=None):
        """Every unmasked spaxel once, colour class by colour class.  Returns the
        number of accepted proposals."""
        accepted = 0
        P = self.params.shape[2]
        for (ys, xs) in self.colours:
            n = len(ys)
            if n == 0:
                continue
            u = self.rng.random((n, P))
            lines = np.empty((n, 2, self.D))
            in3 = np.empty((n, 3))
            p_new_all = np.empty((n, P))
            for i in range(n):
                y, x = int(ys[i]), int(xs[i])
                p_old = self.params[y, x].copy()
                p_new = p_old + self.amp * np.tan(np.pi * (u[i] - 0.5))      # lib/run.py:570-579
                self.model.post_jump(self.run, p_old, p_new)                  # lib/run.py:374
                oob = bool((p_new < self.min_b).any() or (p_new > self.max_b).any())
                lines[i, 0] = self.unit[y, x]
                lines[i, 1] = self.unit_line(p_new)
                in3[i, 0] = p_old[self.g] if self.g is not None else 1.0
                in3[i, 1] = 1.0 if oob else 0.0
                p_new_all[i] = p_new
            in3[:, 2] = np.log(1.0 - self.rng.random(n))                      # log U, U in (0,1]
            out = self.eng.mh_colour_lines(s + self.sweep_origin, ys * self.W + xs, in3, lines,
                                           gibbs=self.g is not None)
            for i in range(n):
                y, x = int(ys[i]), int(xs[i])
                if out[i, 0] != 0.0:
                    self.params[y, x] = p_new_all[i]
                    self.unit[y, x] = lines[i, 1]
                    accepted += 1
                if self.g is not None:
                    self.params[y, x, self.g] = out[i, 1]
                if dlog is not None:
                    dlog[y, x] = out[i, 2]
        if self.refresh_every > 0 and s % self.refresh_every == 0:
            self.refresh()
        return accepted
