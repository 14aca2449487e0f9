# coding=utf-8
"""
``Run`` -- the MCMC deconvolution runner, drop-in for the reference's
``lib/run.py`` class ``Run`` (constructor kwargs :95-109, result attributes
:540-549, helper methods :553-708, 742-840).

The host control flow (validation, setup, stopping rule, chain bookkeeping) is
restated here; everything inside the reference's per-spaxel loop
(lib/run.py:367-519) and the forward models (:597-708, 999-1031) runs on the
GPU through ``libdeconv3d_hip.so``.  There is no CPU fallback.

Differences from the reference, all deliberate:
  * additive kwargs ``seed=`` (the reference is unseeded), ``device=``,
    ``refresh_every=`` (the reference hard-codes 1000, lib/run.py:525);
  * spaxels are scanned colour by colour ((y mod fh, x mod fw) classes, whose
    FSF windows are disjoint) instead of row-major -- the reference declares
    the scan order overridable (lib/run.py:553-560);
  * the (H,W,D,H,W) ``contributions`` array (lib/run.py:285-288) is never
    built: a spaxel's old contribution is recomputed from its parameters;
  * chain / likelihoods are NaN-initialised instead of uninitialised memory;
  * the user's mask is copied, not mutated; zero variances become 1e12 for
    ndarray input too; NaN voxels get zero weight in the Gibbs sums as well;
  * 1-D ``initial_parameters`` are broadcast as the docstring promises;
  * ``SingleGaussianLineModel`` (and subclasses that only change names/bounds)
    is evaluated on the device; any other ``LineModel`` plugin is evaluated on
    the host per colour class (host_model.py) with the same device kernels for
    everything else -- correct, but python-speed.
"""
from __future__ import annotations

import logging
import math
from os.path import splitext

import numpy as np

from . import _lib
from .cube import Cube, read_fits
from .instruments import Instrument
from .line_models import LineModel, SingleGaussianLineModel
from .math_utils import median_clip

logging.basicConfig(level=logging.INFO)
logger = logging.getLogger('deconv3d')

CIRCLE_4TH = np.pi / 2.


class Run:
    """
    Main runner of the deconvolution::

        cube = Cube.from_fits('my_fits.fits')
        inst = MUSE()
        run = Run(cube, inst, max_iterations=10000)
        run.plot_chain()

    Arguments are those of the reference (lib/run.py:51-109): ``cube``
    (FITS path or Cube), ``instrument``, ``mask``, ``variance``, ``model``,
    ``initial_parameters``, ``jump_amplitude``, ``gibbs_apriori_variance``,
    ``max_iterations``, ``keep_one_in``, ``write_every``,
    ``min_acceptance_rate``; plus ``seed``, ``device``, ``refresh_every``,
    ``sweeps_per_call``, ``checkpoint`` (file prefix written every
    ``write_every`` iterations) and ``resume_state`` (the ``<prefix>_state.npz``
    of a checkpoint: the sweep numbering -- hence the random-number streams -- and the
    accepted count of the stopping rule continue where the checkpointed run stopped; pass the
    checkpoint's ``<prefix>_parameters.npy`` as ``initial_parameters``), and
    ``chain_file`` (file prefix: chain and likelihoods live in memory-mapped
    ``<prefix>_chain.npy`` / ``<prefix>_likelihoods.npy`` instead of RAM, written as the
    device streams saved sweeps out -- a 300x300 chain of 50 000 sweeps is 108 GB), and
    ``chains`` (R >= 2 independent chains of the same cube, seeds ``seed + r``, advanced
    TOGETHER: a cube whose colour launches leave the chip idle -- the reference's own fixtures
    are 24x30x21 -- runs R chains in about the time of one, see `chains` below).

    With ``chains=R``: ``run.chains[r]`` / ``run.all_likelihoods[r]`` are chain r's arrays
    (``run.chain`` / ``run.likelihoods`` are chain 0's; chain r is bit for bit the chain of
    ``Run(..., seed=seed + r)``), ``extract_parameters`` pools the chains, ``run.rhat`` is the
    per-parameter Gelman-Rubin map, ``run.acceptance_rates`` the per-chain rates; the stopping
    rule looks at the pooled acceptance rate.  ``initial_parameters`` may be 4-D, one map per
    chain; a checkpoint holds all R chains (``<prefix>_parameters.npy`` is that 4-D array, chain
    r > 0's slots are ``<prefix>_c<r>_chain.npy``) and resumes R chains.  A custom
    (host-evaluated) line model advances its R chains one after the other.
    """

    def __init__(
        self,
        cube,
        instrument=None,
        mask=None,
        variance=None,
        model=SingleGaussianLineModel,
        initial_parameters=None,
        jump_amplitude=0.1,
        gibbs_apriori_variance=None,
        max_iterations=100000,
        keep_one_in=1,
        write_every=10000,
        min_acceptance_rate=0.01,
        seed=12345,
        device=0,
        refresh_every=1000,
        sweeps_per_call=None,
        checkpoint=None,
        resume_state=None,
        chain_file=None,
        chains=1,
    ):
        # lib/run.py:112-114
        assert keep_one_in > 0, "keep_one_in= MUST be a positive integer"
        assert write_every > 0, "write_every= MUST be a positive integer"
        assert max_iterations > 0, "max_iterations= MUST be a positive integer"
        self.logger = logger
        self.keep_one_in = int(keep_one_in)
        self.max_iterations = int(max_iterations)
        self.write_every = int(write_every)
        n_chains = int(chains)
        assert n_chains >= 1, "chains= MUST be a positive integer"
        self.n_chains = n_chains

        # ---- input cube (lib/run.py:119-143) --------------------------------
        if isinstance(cube, str):
            cube = Cube.from_fits(cube)
        if not isinstance(cube, Cube):
            raise TypeError("Provided cube is not a HyperspectralCube")
        if cube.is_empty():
            raise ValueError("Provided cube is empty")
        self.cube = cube
        signal_max = np.nanmax(self.cube.data)
        assert signal_max > 1e-10, \
            "The input cube has data that is too small and will cause " \
            "numerical instability, infinite loops, or worse : bad science."
        cube_shape = cube.data.shape
        cube_depth, cube_height, cube_width = cube_shape

        # ---- mask (lib/run.py:151-165), copied instead of mutated -------------
        if mask is None:
            mask = np.ones((cube_height, cube_width))
        if isinstance(mask, str):
            mask, _ = read_fits(mask)
        mask = np.array(mask, dtype=np.float64)
        if mask.shape != (cube_height, cube_width):
            raise ValueError("Mask MUST have (%d, %d) shape, got %s."
                             % (cube_height, cube_width, str(mask.shape)))
        mask[np.isnan(np.sum(self.cube.data, 0))] = 0
        self.mask = mask
        spaxels_count = int(np.sum(self.mask == 1))
        cnt_iterations = int(math.ceil(max_iterations / float(keep_one_in)))

        # ---- variance (lib/run.py:170-200) ---------------------------------
        if variance is not None:
            if isinstance(variance, str):
                variance = Cube.from_fits(variance)
            if isinstance(variance, Cube):
                if variance.data is None:
                    self.logger.warning("Provided variance cube is empty")
                variance_cube = variance.data
            elif isinstance(variance, np.ndarray):
                variance_cube = variance
            else:
                raise TypeError("Provided variance is not a Cube")
            variance_cube = np.where(variance_cube == 0.0, 1e12, variance_cube)
        else:
            sub_data = np.copy(cube.data[2:-2, 2:-4, 2:4])
            _, clip_sigma, _ = median_clip(sub_data, 2.5)
            if clip_sigma == 0:
                clip_sigma = 1e-20
            variance_cube = np.ones(cube_shape) * clip_sigma ** 2
        if variance_cube.shape != cube_shape:
            raise ValueError("Provided variance has not the correct shape."
                             "Expected %s, got %s" % (str(cube_shape), str(variance_cube.shape)))
        self.variance_cube = variance_cube
        self.error_cube = np.sqrt(self.variance_cube)

        # ---- instrument and taps (lib/run.py:202-224) ------------------------
        if not isinstance(instrument, Instrument):
            raise TypeError("Provided instrument is not an Instrument")
        self.instrument = instrument
        lsf = self.instrument.lsf.as_vector(self.cube)
        self.lsf = None if lsf is None else np.asarray(lsf, dtype=np.float64)
        self.fsf = np.asarray(self.instrument.fsf.as_image(self.cube), dtype=np.float64)
        if self.fsf.ndim != 2 or self.fsf.shape[0] % 2 == 0 or self.fsf.shape[1] % 2 == 0:
            raise ValueError("FSF *must* be of odd dimensions")
        if self.lsf is not None and self.lsf.shape != (cube_depth,):
            raise ValueError("LSF must have the length of the cube's spectral axis (%d), got %s"
                             % (cube_depth, str(self.lsf.shape)))

        # ---- model (lib/run.py:226-265) --------------------------------------
        if isinstance(model, LineModel):
            self.model = model
        else:
            self.model = model()
            if not isinstance(self.model, LineModel):
                raise TypeError("Provided model is not a LineModel")
        self._host_model = not self._model_is_on_device()
        min_boundaries = np.array(self.model.min_boundaries(self), dtype=np.float64)
        max_boundaries = np.array(self.model.max_boundaries(self), dtype=np.float64)
        names = self.model.parameters()
        self.logger.info("Min boundaries : %s" % dict(zip(names, min_boundaries)))
        self.logger.info("Max boundaries : %s" % dict(zip(names, max_boundaries)))
        if not (np.isfinite(min_boundaries).all() and np.isfinite(max_boundaries).all()):
            raise ValueError("Boundaries are not finite: min %s, max %s (NaN or infinite data?)"
                             % (min_boundaries, max_boundaries))
        if (min_boundaries > max_boundaries).any():
            raise ValueError("Boundaries are inconsistent: min > max.")
        parameters_count = len(names)
        jumping_amplitude = np.ones(parameters_count) * np.array(jump_amplitude)
        gpi = self.model.gibbs_parameter_index()
        if gpi is not None:                                   # lib/run.py:255-265
            jumping_amplitude[gpi] = 0
            if gibbs_apriori_variance is None:
                gibbs_apriori_variance = float(max_boundaries[gpi] ** 2)
        elif gibbs_apriori_variance is None:
            gibbs_apriori_variance = 1.0
        self.min_boundaries = min_boundaries
        self.max_boundaries = max_boundaries
        self.jumping_amplitude = jumping_amplitude
        self.gibbs_apriori_variance = gibbs_apriori_variance
        self.seed = int(seed)

        # ---- chain storage (lib/run.py:267-281), NaN instead of garbage ------
        # (one chain array per chain; chain r > 0 of a memory-mapped run lives in <prefix>_c<r>_*)
        self._chain_file = chain_file
        self._likelihoods_maps = []
        self.chains, all_likelihoods = [], []
        try:
            chain_shape = (cnt_iterations, cube_height, cube_width, parameters_count)
            for r in range(n_chains):
                if chain_file is not None:
                    # pages are created as saved sweeps arrive; slots never written (early
                    # stop) are NaN-filled at the end like the in-memory chain
                    prefix = chain_file if r == 0 else "%s_c%d" % (chain_file, r)
                    ch = np.lib.format.open_memmap(
                        "%s_chain.npy" % prefix, mode="w+", dtype=np.float64, shape=chain_shape)
                    lk = np.lib.format.open_memmap(
                        "%s_likelihoods.npy" % prefix, mode="w+", dtype=np.float64,
                        shape=chain_shape[:3])
                    lk[0] = np.nan
                    self._likelihoods_maps.append(lk)
                else:
                    ch = np.full(chain_shape, np.nan)
                    lk = np.full(chain_shape[:3], np.nan)
                self.chains.append(ch)
                all_likelihoods.append(lk)
        except MemoryError:
            self.logger.error("Not enough RAM available for that many iterations. "
                              "Use a higher value in the keep_one_in= parameter.")
            raise
        self.chain = self.chains[0]
        likelihoods = all_likelihoods[0]
        self._likelihoods_map = self._likelihoods_maps[0] if self._likelihoods_maps else None

        # ---- initial parameters (lib/run.py:293-314) -------------------------
        # (chain r draws its start from the generator of seed + r: it is the chain of
        # Run(..., seed=seed + r))
        if initial_parameters is not None:
            if isinstance(initial_parameters, str):
                initial_parameters = np.load(initial_parameters)
            initial_parameters = np.array(initial_parameters, dtype=np.float64)
            if initial_parameters.ndim == 1:
                if initial_parameters.shape[0] != parameters_count:
                    raise ValueError("1D initial params MUST have %d values, got %d."
                                     % (parameters_count, initial_parameters.shape[0]))
                initial_parameters = np.tile(initial_parameters, (cube_height, cube_width, 1))
            per_chain = initial_parameters.ndim == 4
            if per_chain and initial_parameters.shape[0] != n_chains:
                raise ValueError("4D initial params MUST hold one map per chain (%d), got %d."
                                 % (n_chains, initial_parameters.shape[0]))
            ip_shape = initial_parameters.shape[1:] if per_chain else initial_parameters.shape
            if ip_shape[0] != cube_height or ip_shape[1] != cube_width:
                raise ValueError("Initial params MUST have (%d, %d) shape, got (%d, %d)."
                                 % (cube_height, cube_width, ip_shape[0], ip_shape[1]))
            for r in range(n_chains):
                self.chains[r][0] = initial_parameters[r] if per_chain else initial_parameters
        else:
            for r in range(n_chains):
                draws = np.random.default_rng(self.seed + r).random(
                    (cube_height, cube_width, parameters_count))
                self.chains[r][0] = min_boundaries + (max_boundaries - min_boundaries) * draws

        # a resumed run continues the checkpointed run's sweep numbering: sweep s of
        # this segment draws the random numbers of sweep s + origin
        self.sweep_origin = 0
        resumed_accepted = None
        resumed_per_chain = None
        if resume_state is not None:
            state = np.load(resume_state) if isinstance(resume_state, str) else resume_state
            files = getattr(state, "files", state)
            saved_chains = int(state["n_chains"]) if "n_chains" in files else 1
            if saved_chains != n_chains:
                raise ValueError("resume_state holds %d chain(s), this run has chains=%d"
                                 % (saved_chains, n_chains))
            if int(state["seed"]) != self.seed:
                self.logger.warning("resume_state was written with seed %d, this run uses %d"
                                    % (int(state["seed"]), self.seed))
            self.sweep_origin = int(state["sweep_origin"]) + int(state["iteration"]) - 1
            # totals over every earlier segment (older checkpoints hold one segment's)
            resumed_accepted = (
                int(state["total_accepted"] if "total_accepted" in files else state["accepted_count"]),
                int(state["total_iterations"] if "total_iterations" in files else state["iteration"]))
            if n_chains > 1 and "per_chain_accepted" in files:
                resumed_per_chain = [int(v) for v in state["per_chain_accepted"]]

        # ---- device context ----------------------------------------------
        self.engines = []
        for r in range(n_chains):
            eng = _lib.Engine(cube_shape, self.fsf.shape, device=device)
            self.engines.append(eng)
            eng.set_taps(self.fsf, self.lsf)
            eng.set_data(self.cube.data, self.variance_cube, mask=self.mask)
        self.engine = self.engines[0]
        host_chain = None
        if self._host_model:
            from .host_model import HostModelChain
            self.logger.info("Line model %s is evaluated on the host (slower path)."
                             % type(self.model).__name__)
            # (chains=R: one host chain per engine, seeds seed + r, advanced one after the other)
            self._host_chains = [
                HostModelChain(self, self.engines[r], self.chains[r][0], min_boundaries,
                               max_boundaries, jumping_amplitude, gibbs_apriori_variance,
                               self.seed + r, refresh_every) for r in range(n_chains)]
            host_chain = self._host_chain = self._host_chains[0]
        else:
            for r, eng in enumerate(self.engines):
                eng.set_params(self.chains[r][0])
                eng.mh_config(min_boundaries, max_boundaries, jumping_amplitude,
                              gibbs_apriori_variance, seed=self.seed + r,
                              refresh_every=refresh_every)
        if resume_state is not None:
            if host_chain is None:
                for eng in self.engines:
                    eng.set_sweep_origin(self.sweep_origin)
            else:
                for hc in self._host_chains:
                    hc.set_sweep_origin(self.sweep_origin)
        self.logger.info("Iteration #1")
        if host_chain is None:
            for eng in self.engines:
                eng.residual(fetch=False)              # lib/run.py:317-334

        # ---- MH within Gibbs loop (lib/run.py:336-537) -------------------------
        cur_iteration = 1
        cur_acceptance_rate = 0.
        accepted_count = spaxels_count * n_chains  # first iteration counts as accepted
        per_chain_accepted = [spaxels_count] * n_chains
        # (accepted, iterations) of earlier segments: the running acceptance rate of the
        # stopping rule (lib/run.py:344-359) is that of the WHOLE chain.  This segment's
        # iteration 1 is the resumed state itself, already counted there.
        self._resumed_from = resumed_accepted
        import time as _time
        self.mh_seconds = 0.0                      # wall time of the device calls of the loop
        self._acc_base = resumed_accepted[0] - spaxels_count * n_chains if resumed_accepted else 0
        self._it_base = resumed_accepted[1] - 1 if resumed_accepted else 0
        # (per chain: what the checkpoint recorded, else an even share of the total)
        if resumed_accepted and resumed_per_chain is None:
            resumed_per_chain = [resumed_accepted[0] // n_chains] * n_chains
        self._acc_base_chain = [a - spaxels_count for a in resumed_per_chain] if resumed_accepted \
            else [0] * n_chains
        # the reference re-evaluates the stopping rule (and logs) every sweep
        # (lib/run.py:344-364): one sweep per device call whenever the rule is
        # armed, so that the run stops exactly where the reference would; with
        # min_acceptance_rate <= 0 sweeps are batched up to the next saved one
        if sweeps_per_call is None:
            sweeps_per_call = 1 if min_acceptance_rate > 0 else max(1, min(int(keep_one_in), 64))
        if host_chain is not None:
            sweeps_per_call = 1
        self.iterations_done = 1
        while cur_iteration < max_iterations and \
                (cur_acceptance_rate > min_acceptance_rate or cur_acceptance_rate == 0.):
            max_accepted_count = spaxels_count * n_chains * (cur_iteration + self._it_base)
            if max_accepted_count > 0:
                cur_acceptance_rate = float(accepted_count + self._acc_base) / float(max_accepted_count)
            n = min(sweeps_per_call, max_iterations - cur_iteration)
            self.logger.info("Iteration #%d / %d, %2.0f%%" %
                             (cur_iteration + 1, max_iterations, 100 * cur_acceptance_rate))
            if host_chain is not None:
                save = cur_iteration % keep_one_in == 0
                slot = cur_iteration // keep_one_in
                acc = [hc.sweep(cur_iteration, all_likelihoods[r][slot] if save else None)
                       for r, hc in enumerate(self._host_chains)]
                accepted_count += sum(acc)
                per_chain_accepted = [a + b for a, b in zip(per_chain_accepted, acc)]
                if save:
                    for r, hc in enumerate(self._host_chains):
                        self.chains[r][slot] = hc.params
            elif n_chains == 1:
                t_call = _time.perf_counter()
                accepted_count += self.engine.mh_sweeps(n, cur_iteration, keep_one_in,
                                                        self.chain, likelihoods)
                self.mh_seconds += _time.perf_counter() - t_call
            else:
                t_call = _time.perf_counter()
                acc = self._sweep_chains(n, cur_iteration, all_likelihoods)
                self.mh_seconds += _time.perf_counter() - t_call
                accepted_count += sum(acc)
                per_chain_accepted = [a + b for a, b in zip(per_chain_accepted, acc)]
            before = cur_iteration
            cur_iteration += n
            # write_every: documented (lib/run.py:89-92) but never used by the
            # reference; here it is the checkpoint cadence when a path is given
            if checkpoint is not None and before // write_every != cur_iteration // write_every:
                self.iterations_done = cur_iteration
                self._write_checkpoint(checkpoint, cur_iteration, accepted_count, per_chain_accepted)
        self.iterations_done = cur_iteration
        self.acceptance_rate = float(accepted_count + self._acc_base) / \
            float(max(spaxels_count * n_chains * (cur_iteration + self._it_base), 1))
        if n_chains > 1:
            self.acceptance_rates = [(a + b) / float(max(spaxels_count * (cur_iteration + self._it_base), 1))
                                     for a, b in zip(per_chain_accepted, self._acc_base_chain)]
        else:
            self.acceptance_rates = [self.acceptance_rate]
        if chain_file is not None:
            n_valid = (cur_iteration - 1) // self.keep_one_in + 1
            for ch, lk in zip(self.chains, all_likelihoods):
                ch[n_valid:] = np.nan
                lk[n_valid:] = np.nan
                ch.flush()
                lk.flush()

        # ---- outputs (lib/run.py:539-549) ------------------------------------
        self.likelihoods = likelihoods
        self.all_likelihoods = all_likelihoods
        self.rhat = self.gelman_rubin() if n_chains > 1 else None
        self.parameters = self.extract_parameters()
        self.convolved_cube = Cube(data=self.simulate_convolved(cube_shape, self.parameters),
                                   meta=self.cube.meta, x=cube.x, y=cube.y, z=cube.z)
        self.clean_cube = Cube(data=self.simulate_clean(cube_shape, self.parameters),
                               meta=self.cube.meta, x=cube.x, y=cube.y, z=cube.z)

    # ------------------------------------------------------------------------

    def _model_is_on_device(self):
        """The HIP kernels evaluate SingleGaussianLineModel themselves (subclasses
        may change names/bounds but not the curve or the Gibbs index); any other
        LineModel plugin is evaluated on the host (host_model.HostModelChain):
        same device kernels for LSF, window statistics, accept, Gibbs draw and
        residual, but python-speed proposals and modelize() calls."""
        m = self.model
        return (isinstance(m, SingleGaussianLineModel)
                and type(m).modelize is SingleGaussianLineModel.modelize
                and type(m).gaussian is SingleGaussianLineModel.gaussian
                and type(m).post_jump is LineModel.post_jump
                and m.gibbs_parameter_index() == 0
                and len(m.parameters()) == 3)

    def _sweep_chains(self, n, first, all_likelihoods):
        """n sweeps of every chain: ONE launch per colour class for all of them where the
        library can (d3d_mh_sweeps_batch: cubes up to 256 channels -- a small cube's launch
        carries R times the windows for about the same latency), else the chains' own
        launches on their own streams, enqueued concurrently (ensemble.sweep_chains)."""
        from . import ensemble
        if getattr(self, "_batched", True):
            try:
                acc = ensemble.sweep_chains_batched(self.engines, n, first, self.keep_one_in,
                                                    self.chains, all_likelihoods)
                self._batched = True
                return acc
            except NotImplementedError as exc:
                self.logger.info("chains advance on their own streams (%s)" % exc)
                self._batched = False
        return ensemble.sweep_chains(self.engines, n, first, self.keep_one_in, self.chains,
                                     all_likelihoods)

    def gelman_rubin(self, percentage=50.):
        """Per-parameter potential scale reduction R-hat (Gelman & Rubin 1992) over the last
        ``percentage`` % of the saved samples of the ``chains=R`` chains, shape (H, W, P):
        sqrt(((n-1)/n W + B/n) / W), W the mean within-chain variance and B/n the variance of
        the chain means.  NaN where a parameter did not move (masked spaxels)."""
        if self.n_chains < 2:
            raise ValueError("R-hat needs chains >= 2")
        n_valid = min((self.iterations_done - 1) // self.keep_one_in + 1, self.chain.shape[0])
        if n_valid < 2:                          # (a variance needs two saved samples)
            return np.full(self.chain.shape[1:], np.nan)
        first = min(int((100. - percentage) * n_valid / 100.), n_valid - 2)
        tail = np.stack([np.asarray(ch[first:n_valid]) for ch in self.chains])   # (R, n, H, W, P)
        n = tail.shape[1]
        with np.errstate(invalid="ignore", divide="ignore"):
            within = tail.var(axis=1, ddof=1).mean(axis=0)
            between = tail.mean(axis=1).var(axis=0, ddof=1)
            return np.sqrt(((n - 1.) / n * within + between) / within)

    def _write_checkpoint(self, name, iteration, accepted_count, per_chain_accepted=None):
        """`<name>_parameters.npy` (current map, reusable as initial_parameters,
        lib/run.py:790-797), `<name>_chain.npy` (slots written so far) and
        `<name>_state.npz` (iteration, seed, accepted count, `n_valid` = chain slots
        written so far: `resume_state=`).
        A memory-mapped chain (`chain_file=`) is flushed where it lives instead of
        copied: the checkpoint then names its files, and `chain_file == checkpoint`
        cannot rewrite the file under the open mapping.
        ``chains=R``: the parameter file holds the R maps, (R, H, W, P) -- what
        `initial_parameters=` takes for R chains --, chain r > 0 goes to
        `<name>_c<r>_chain.npy`, and the state records the chains' own accepted counts."""
        n_valid = (iteration - 1) // self.keep_one_in + 1
        state = dict(iteration=iteration, seed=self.seed,
                     accepted_count=accepted_count, sweep_origin=self.sweep_origin,
                     keep_one_in=self.keep_one_in,
                     total_accepted=accepted_count + self._acc_base,
                     total_iterations=iteration + self._it_base,
                     n_valid=n_valid, n_chains=self.n_chains,
                     chain_file="" if self._chain_file is None else str(self._chain_file))
        if self.n_chains > 1:
            state["per_chain_accepted"] = np.array(
                [a + b for a, b in zip(per_chain_accepted, self._acc_base_chain)], dtype=np.int64)
        np.savez("%s_state.npz" % name, **state)
        if self._host_model:
            maps = [hc.params for hc in self._host_chains]
        else:
            maps = [eng.get_params() for eng in self.engines]
        params = maps[0] if self.n_chains == 1 else np.stack(maps)
        np.save("%s_parameters.npy" % name, params)
        if self._chain_file is not None:
            # slots past n_valid (recorded in the state file) are not written yet
            for ch, lk in zip(self.chains, self._likelihoods_maps):
                ch.flush()
                lk.flush()
        else:
            for r, ch in enumerate(self.chains):
                prefix = name if r == 0 else "%s_c%d" % (name, r)
                np.save("%s_chain.npy" % prefix, ch[:n_valid])
        self.logger.info("checkpoint at iteration %d (%d accepted) -> %s_*.npy"
                         % (iteration, accepted_count, name))

    # ITERATORS ###############################################################

    def spaxel_iterator(self):
        """(y, x) of every unmasked spaxel, row-major (lib/run.py:553-566).
        The device sweep visits the same set colour by colour."""
        h, w = self.mask.shape
        for y in range(h):
            for x in range(w):
                if self.mask[y, x] == 1:
                    yield (y, x)

    # MCMC ####################################################################

    def jump_from(self, parameters, amplitude):
        """Host mirror of the Cauchy proposal (lib/run.py:570-579); the device
        draws its own from Philox."""
        size = len(parameters)
        u = np.random.default_rng().uniform(-CIRCLE_4TH, CIRCLE_4TH, size=size)
        return parameters + amplitude * np.tan(u)

    def extract_parameters(self, percentage=20.):
        """Mean of the last ``percentage`` % of the saved chain -- of every chain with
        ``chains=R`` -- (lib/run.py:581-593); slots never written (early stop) are ignored."""
        n_valid = (self.iterations_done - 1) // self.keep_one_in + 1
        n_valid = max(1, min(n_valid, self.chain.shape[0]))
        s = int((100. - percentage) * n_valid / 100.)
        if self.n_chains == 1:
            return np.nanmean(self.chain[s:n_valid, ...], 0)
        # chains=R: the chains are pooled (equal lengths: the mean of their means)
        return np.nanmean(np.stack([np.nanmean(ch[s:n_valid, ...], 0) for ch in self.chains]), 0)

    # SIMULATOR ###############################################################

    def simulate_clean(self, shape, parameters):
        """Cube of the raw lines (lib/run.py:597-621), built on the device."""
        self._check_shape(shape)
        if self._host_model:
            return self._host_chain.clean_cube(np.asarray(parameters, dtype=np.float64))
        return self.engine.simulate(parameters, convolved=False)

    def simulate_convolved(self, shape, parameters):
        """Cube of the LSF- and FSF-convolved lines (lib/run.py:623-652), by the
        fused device forward model."""
        self._check_shape(shape)
        if self._host_model:
            return self.engine.convolve(
                self._host_chain.clean_cube(np.asarray(parameters, dtype=np.float64)))
        return self.engine.simulate(parameters, convolved=True)

    def contribution_of_spaxel(self, x, y, parameters, cube_width, cube_height, cube_depth,
                               fsf=None, lsf=None, lsf_fft=None):
        """
        Full-size cube holding the convolved contribution of one spaxel's line
        (lib/run.py:654-708).  Returns ``(cube, lsf_fft)`` like the reference;
        ``lsf_fft`` is passed through (the device does not use FFTs).  The taps
        are the run's own.
        """
        self._check_shape((cube_depth, cube_height, cube_width))
        # the line on the host (one spectrum), LSF and FSF on the device; like the
        # reference this ignores the mask, and the chain state is not touched
        one = np.zeros((cube_depth, cube_height, cube_width))
        one[:, y, x] = np.asarray(self.model.modelize(self, np.arange(cube_depth, dtype=float),
                                                      parameters), dtype=np.float64)
        return self.engine.convolve(one), lsf_fft

    def _check_shape(self, shape):
        if tuple(shape) != tuple(self.cube.data.shape):
            raise ValueError("shape %s differs from the run's cube %s"
                             % (tuple(shape), tuple(self.cube.data.shape)))

    # SAVES ###################################################################

    def save(self, name, clobber=False):
        """Write ``<name>_parameters.npy``, ``_chain.npy``, ``_matlab.mat``,
        ``_images.png``, ``_chain.png``, ``_convolved_cube.fits``,
        ``_clean_cube.fits``, ``_result.npz`` (lib/run.py:742-788)."""
        self.save_parameters_npy("%s_parameters.npy" % name)
        self.save_chain_npy("%s_chain.npy" % name)
        try:
            self.save_matlab("%s_matlab.mat" % name)
        except Exception as e:  # noqa
            self.logger.error(str(e))
        self.plot_images("%s_images.png" % name)
        self.plot_chain(filepath="%s_chain.png" % name)
        self.convolved_cube.to_fits("%s_convolved_cube.fits" % name, clobber)
        self.clean_cube.to_fits("%s_clean_cube.fits" % name, clobber)
        np.savez("%s_result.npz" % name, chain=self.chain, likelihoods=self.likelihoods,
                 fsf=self.fsf, lsf=self.lsf if self.lsf is not None else np.zeros(0))

    def save_parameters_npy(self, filepath):
        """lib/run.py:790-797; reusable as ``initial_parameters``."""
        np.save(filepath, self.extract_parameters())

    def save_chain_npy(self, filepath):
        """lib/run.py:799-810."""
        np.save(filepath, self.chain)

    def save_matlab(self, filepath):
        """lib/run.py:812-840."""
        from scipy.io import savemat
        savemat(filepath, dict(parameters=self.extract_parameters(), chain=self.chain))

    # PLOTS ###################################################################

    def plot_chain(self, x=None, y=None, filepath=None, bound=True):
        """Chain and log acceptance ratio of one spaxel (lib/run.py:844-894)."""
        from matplotlib import pyplot as plot
        self._check_image_filepath(filepath)
        if x is None:
            x = int(math.floor(self.cube.shape[2] / 2.))
        if y is None:
            y = int(math.floor(self.cube.shape[1] / 2.))
        chain_t = np.transpose(self.chain[:, y, x, :])
        names = self.model.parameters()
        bmin, bmax = self.min_boundaries, self.max_boundaries
        plot.clf()
        for i, name in enumerate(names):
            plot.subplot2grid((2, len(names)), (0, i))
            plot.plot(chain_t[i])
            if bound:
                plot.ylim(bmin[i], bmax[i])
            plot.title(name, fontsize='small')
        plot.subplot2grid((2, len(names)), (1, 0), colspan=len(names))
        plot.plot(self.likelihoods[:, y, x])
        plot.title('likelihood', fontsize='small')
        if filepath is None:
            plot.show()
        else:
            plot.savefig(filepath)

    def plot_images(self, filepath=None):
        """Mosaic of measured / convolved / FSF / clean / mask images
        (lib/run.py:896-985)."""
        from matplotlib import pyplot as plot
        self._check_image_filepath(filepath)
        p = self.extract_parameters()
        panels = [
            ('Measured', np.nanmean(self.cube.data, 0)),
            ('Simulation Convolved', self.simulate_convolved(self.cube.data.shape, p).mean(0)),
            ('FSF', self.fsf),
            ('Simulation Clean', self.simulate_clean(self.cube.data.shape, p).mean(0)),
            ('Mask', self.mask),
        ]
        fig = plot.figure(1, figsize=(16, 9))
        plot.clf()
        plot.subplots_adjust(wspace=0.25, hspace=0.25, bottom=0.05, top=0.95, left=0.05,
                             right=0.95)
        for i, (title, image) in enumerate(panels):
            sub = fig.add_subplot(2, 3, i + 1)
            sub.set_title(title)
            plot.imshow(image, interpolation='nearest', origin='lower')
            plot.xticks(fontsize=8)
            plot.yticks(fontsize=8)
            plot.colorbar().ax.tick_params(labelsize=8)
        if filepath is None:
            plot.show()
        else:
            plot.savefig(filepath)

    def _check_image_filepath(self, filepath):
        if filepath is not None:
            _, extension = splitext(filepath)
            supported = ['.png', '.pdf']
            if extension not in supported:
                raise ValueError("Extension '%s' is not supported, you may use one of %s"
                                 % (extension, ', '.join(supported)))
