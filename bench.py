#!/usr/bin/env python
# coding=utf-8
"""
bench.py -- spaxel-updates/sec of the MH-within-Gibbs hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one full sweep (every unmasked spaxel updated once: proposal,
windowed 1/2 chi2 old/new, accept, Gibbs amplitude draw, residual write-back)
over BASELINE config 3: a synthetic MUSE-WFM-sized 300x300x128 cube, Moffat
11x11 FSF, MUSE-like LSF, fp64, inputs resident in HBM before the timed region.
N > 1 runs N independent chains, one per GPU (BASELINE config 5, "weak"
scaling, no data-path collective); `--mode tiled` runs ONE chain spatially
tiled over the ranks with halo exchange (config 4).

Launch: under a launcher (RANK / WORLD_SIZE in the environment, e.g. `python -m
torch.distributed.run --nproc-per-node N bench.py --gpus N ...`) every process is
one rank.  Without one, `--gpus N` with N > 1 starts the N ranks itself (a child
`torch.distributed.run` on 127.0.0.1, started before this process touches the
GPU) and exits non-zero when fewer than N devices are visible.

Prints ONE JSON line (rank 0).  Besides the contract keys it carries
  roofline             -- the MH sweep kernel (k_mh_ws): algorithmic bytes / launch time
  roofline_beyond_mall -- the same kernel on a 300x300x256 cube, whose residual +
                          1/variance (369 MB) exceed the 256 MB Infinity Cache
  roofline_deep        -- the sweep on a 200x200x1024 cube: the z-blocked form of the kernel
                          (one workgroup per window and 256-channel block + a decision launch)
  chains_batched       -- config 2's cube as 16 independent chains in one launch per colour class
  roofline_conv        -- the separable LSF (x) FSF convolution of one cube
  roofline_forward     -- parameters -> convolved cube (line build + convolution)
  roofline_chi2        -- the per-spaxel chi2 reduction of the residual
  cpu_baseline         -- the oracle's memory-sane numpy update loop on the host cores
  cpu_baseline_conv    -- the oracle's LSF (x) FSF convolution of one cube on the host
  host                 -- CPU model string and core counts of the box
(the CPU baselines and the extra legs at N = 1 only), and at N > 1
  config4_tiled        -- ONE chain of the same cube tiled over the same N GPUs (row strips,
                          halo rectangles by RCCL inside the library), strong scaling:
                          measured by child processes after the ensemble measurement, so a
                          failure there costs an "error" note, never the contract line
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_VEC_PEAK_TF = 78.6     # vector fp64 peak

WORKLOADS = {
    # name: (D, H, W, fsf size)
    "c3_300x300x128": (128, 300, 300, 11),
    "c2_64x64x64": (64, 64, 64, 11),
    "c1_32x16x16": (32, 16, 16, 9),
    # config 3's footprint and taps at other depths (the convolution kernels' other forms)
    "d64_300x300x64": (64, 300, 300, 11),
    "d256_300x300x256": (256, 300, 300, 11),
    # the 512-thread form of the sweep kernel, and its z-blocked form (a MUSE cube's depth class)
    "d512_300x300x512": (512, 300, 300, 11),
    "d1024_300x300x1024": (1024, 300, 300, 11),
}


def build_taps(D, fsf_size):
    """Taps of SURVEY.md 8(d); formulas of lib/spread_functions.py:165-189
    (Moffat, beta 2.5, FWHM 3 px, evaluated on an fsf_size^2 grid and fed
    through ImageFieldSpreadFunction semantics) and a MUSE-like LSF stand-in
    (box 1 px (x) Gaussian sigma 0.9 px)."""
    from deconv3d_amd.spread_functions import moffat_image, muse_like_lsf_vector
    from deconv3d_amd.spread_functions import gaussian_image, gaussian_lsf_vector_px
    if fsf_size == 9:
        return gaussian_image(3.0), gaussian_lsf_vector_px(D, 0.9088)
    fsf = moffat_image((fsf_size, fsf_size), beta=2.5, fwhm_px=3.0)
    lsf = muse_like_lsf_vector(D, sigma_px=0.9, box_px=1.0)
    return fsf, lsf


def synthetic_inputs(eng, D, H, W, fsf, seed, A0=10.0):
    """SURVEY.md 8(d) synthetic cube; the noiseless model comes from the
    (parity-tested) device forward model so that setup stays fast."""
    rng = np.random.default_rng(seed)
    y, x = np.indices((H, W))
    r2 = (y - H / 2.) ** 2 + (x - W / 2.) ** 2
    truth = np.dstack((A0 * np.exp(-r2 / (2. * (H / 6.) ** 2)),
                       D / 2. + (D / 8.) * np.tanh((x - W / 2.) / (W / 8.)),
                       rng.uniform(1.5, 3.0, size=(H, W))))
    eng.set_params(truth)
    clean = eng.forward()
    sigma = 0.05 * A0 * np.max(fsf)
    # a genuine per-voxel variance cube (as MUSE's STAT extension is): the timed
    # path is the general kernel that streams 1/var, not the uniform-variance
    # variant (reported separately under "uniform_variance")
    var = sigma ** 2 * rng.uniform(0.75, 1.25, size=(D, H, W))
    data = clean + rng.normal(size=(D, H, W)) * np.sqrt(var)
    min_b = np.array([0., 0., 0.])
    max_b = np.array([np.amax(data) / np.amax(fsf), D - 1., float(D)])  # lib/line_models.py:79-90
    init = min_b + (max_b - min_b) * rng.random((H, W, 3))            # lib/run.py:310-314
    return data, var, truth, init, min_b, max_b


def window_voxels(H, W, fh, fw):
    """Sum over spaxels of the clipped window area (lib/run.py:407-410)."""
    fhh, fhw = (fh - 1) // 2, (fw - 1) // 2
    ny = np.minimum(np.arange(H) + fhh + 1, H) - np.maximum(np.arange(H) - fhh, 0)
    nx = np.minimum(np.arange(W) + fhw + 1, W) - np.maximum(np.arange(W) - fhw, 0)
    return int(ny.sum()) * int(nx.sum())


def cpu_baseline(data, var, mask, fsf, lsf, params, min_b, max_b, err, budget_s):
    """The oracle's memory-sane numpy update loop (oracle.mh_update, i.e.
    lib/run.py:369-519 with local windows) on a time-bounded sample of the same
    workload.  Single python process, one core."""
    from oracle import deconv3d_oracle as O
    st = O.MHState.__new__(O.MHState)
    st.data, st.var, st.mask, st.fsf, st.lsf = data, var, mask, fsf, lsf
    st.params = np.array(params)
    st.min_b, st.max_b = min_b, max_b
    st.amp = np.array([0., 0.1, 0.1])
    st.ra = float(max_b[0] ** 2)
    st.seed = 12345
    st.origin = (0, 0, data.shape[2])
    st.last = None
    st.err = np.array(err)
    st.accepted = 0
    st.dlog = np.zeros(mask.shape)
    n = 0
    t0 = time.perf_counter()
    for (y, x) in O.colour_order(mask, *fsf.shape):
        O.mh_update(st, y, x, 1)
        n += 1
        if n >= 200 and time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return n / dt, n, dt


def cpu_baseline_faithful(budget_s):
    """BASELINE.md B-ref: the reference-faithful update (full-cube temporaries
    and the (H,W,D,H,W) contributions array, lib/run.py:285-288, 367-519) on
    config 1 (32x16x16, Gaussian 9x9) -- the only BASELINE shape where that array
    (16.8 MB) is harmless; at 300x300x128 it would be 8.3 TB."""
    from oracle import deconv3d_oracle as O
    D, H, W = 32, 16, 16
    fsf = O.gaussian_fsf_image(3.0)
    lsf = O.gaussian_lsf_vector(D, 0.9088)
    data, var, mask, truth, init, mn, mx = O.synthetic_case(D, H, W, fsf, lsf)
    st = O.RefFaithfulState(data, var, mask, fsf, lsf, init, mn, mx)
    n = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        for (y, x) in O.spaxel_iterator(mask):          # the reference's row-major scan
            O.ref_faithful_update(st, y, x, 1 + n // (H * W))
            n += 1
    dt = time.perf_counter() - t0
    return n / dt, n, dt


def cpu_baseline_conv(data, fsf, lsf, budget_s):
    """the reference's full separable convolution
    (_compute_error_in_one_step, lib/run.py:999-1031: convolve_1d of every
    spectrum, then scipy convolve2d 'same' of every channel) as the oracle
    restates it, on one host core.  The spectral loop runs over a bounded sample
    of spaxels and is scaled to the cube; the spatial pass runs on every channel
    until the budget is spent (then scaled too)."""
    from oracle import deconv3d_oracle as O
    D, H, W = data.shape
    n_sp = min(H * W, 9000)
    t0 = time.perf_counter()
    tmp = np.empty((D, n_sp))
    flat = data.reshape(D, H * W)
    for i in range(n_sp):
        tmp[:, i] = O.spectral_convolve(flat[:, i], lsf)
    t_spec = (time.perf_counter() - t0) * (H * W / float(n_sp))
    t0 = time.perf_counter()
    nz = 0
    for z in range(D):
        O.spatial_convolve(data[z:z + 1], fsf)
        nz += 1
        if time.perf_counter() - t0 > budget_s:
            break
    t_spat = (time.perf_counter() - t0) * (D / float(nz))
    return t_spec + t_spat, t_spec, t_spat, n_sp, nz


def host_info():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"cpu_model": model, "os_cpu_count": os.cpu_count(),
            "sched_affinity": len(os.sched_getaffinity(0))}


def spawn_ranks(args):
    """`--gpus N` without a launcher: start N ranks (one per GPU) as children of a
    torch.distributed.run agent and relay rank 0's JSON line.  Nothing here touches
    the GPU (torch.cuda.device_count() does not initialise it on this image)."""
    import socket
    import subprocess
    import torch
    ndev = torch.cuda.device_count()
    if args.dry_run:
        ndev = max(ndev, args.gpus)
    if args.backend == "nccl" and ndev < args.gpus:
        sys.stderr.write("bench.py: --gpus %d but only %d HIP device(s) visible "
                         "(--backend gloo rehearses the ranks on fewer devices)\n"
                         % (args.gpus, ndev))
        return 2
    if ndev < 1:
        sys.stderr.write("bench.py: no HIP device visible; there is no CPU fallback\n")
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


TILED_LEG_PORT_OFFSET = 173
TILED_LEG_TIMEOUT_S = 240


def start_tiled_leg(args, rank):
    """The config-4 leg of a multi-GPU run: ONE chain tiled over the same N GPUs (halo
    rectangles by RCCL inside the library), measured beside the ensemble `value`.  It
    runs in child processes so that nothing it does can cost the contract line: every
    rank starts its child HERE, before this process touches the GPU; the child imports
    and then waits on its stdin until the ensemble measurement is over."""
    import subprocess
    import tempfile
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    env["MASTER_PORT"] = str(int(env.get("MASTER_PORT", "29500")) + TILED_LEG_PORT_OFFSET)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # (dmabuf IPC: RCCL between processes needs it here)
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(args.gpus), "--mode", "tiled",
           "--steps", str(min(args.steps, 50)), "--warmup", str(min(max(args.warmup, 1), 3)),
           "--workload", args.workload, "--backend", args.backend, "--wait-stdin"]
    if args.tiles:
        cmd += ["--tiles", args.tiles]
    errf = tempfile.TemporaryFile()
    proc = subprocess.Popen(cmd, env=env, stdin=subprocess.PIPE,
                            stdout=subprocess.PIPE if rank == 0 else subprocess.DEVNULL,
                            stderr=errf)
    return proc, errf


def finish_tiled_leg(leg, rank):
    """Release the child, wait for it (bounded), return rank 0's record or an error note."""
    import subprocess
    proc, errf = leg
    try:
        stdout, _ = proc.communicate(b"go\n", timeout=TILED_LEG_TIMEOUT_S)
    except subprocess.TimeoutExpired:
        proc.kill()
        proc.communicate()
        return {"error": "tiled leg timed out after %d s" % TILED_LEG_TIMEOUT_S}
    if rank != 0:
        return None
    for line in reversed((stdout or b"").decode(errors="replace").splitlines()):
        if line.startswith("{"):
            try:
                rec = json.loads(line)
            except ValueError:
                continue
            return {"value": rec["value"], "unit": rec["unit"], "ms_per_step": rec["ms_per_step"],
                    "n_gpus": rec["n_gpus"], "steps": rec["steps"], "scaling": "strong",
                    "acceptance": rec.get("acceptance"), "config": rec["config"],
                    # the run checks itself (tiling.bench_tiled): gathered parameters against
                    # one context given the same parts, on rank 0
                    "bit_identical": rec.get("bit_identical"),
                    "max_abs_param_diff": rec.get("max_abs_param_diff"), "tiles": rec.get("tiles"),
                    "phases_per_sweep": rec.get("phases_per_sweep"),
                    "rccl_ranks": rec.get("rccl_ranks"),
                    "halo_ms_per_sweep": rec.get("halo_ms_per_sweep"),
                    "verified_against": rec.get("verified_against"),
                    # compute side: sum over the phases of the slowest rank's own time
                    "projected_critical_path_ms": rec.get("projected_critical_path_ms"),
                    "slowest_rank": rec.get("slowest_rank"),
                    "critical_path_by_phase": rec.get("critical_path_by_phase")}
    errf.seek(0)
    tail = errf.read().decode(errors="replace").strip().splitlines()[-3:]
    return {"error": "tiled leg exited with code %s" % proc.returncode, "stderr_tail": tail}


def measured_traffic(kernel_prefix, workload, kernel_suffix=""):
    """HBM bytes per launch from the committed rocprofv3 PMC passes
    (tools/profile_round.sh -> profiles/<tag>_traffic.json: 2 x FETCH_SIZE KiB
    [gfx950 correction] + WRITE_SIZE KiB, separate passes).  None if no profile on
    record is for this workload; the string "stale" if the newest one was taken with
    another build of the library than the one loaded now (the JSON carries
    d3d_source_hash()): a traffic figure only counts for the kernels it was measured on."""
    from deconv3d_amd import _lib
    best = None
    pdir = os.path.join(ROOT, "profiles")
    if not os.path.isdir(pdir):
        return None
    for name in sorted(os.listdir(pdir)):
        if "_traffic" not in name or not name.endswith(".json"):     # r04_traffic.json, r04_traffic_conv600.json
            continue
        try:
            rec = json.load(open(os.path.join(pdir, name)))
        except ValueError:
            continue
        if rec.get("workload") != workload:
            continue
        # (the launch-weighted mean over the kernel's pending-layer variants, "..., *, ...", where
        # the summary has one: tools/summarize_profile.py)
        hits = [(k, v) for k, v in rec.get("hbm_bytes_per_launch", {}).items()
                if k.split("::")[-1].startswith(kernel_prefix) and k.endswith(kernel_suffix)]
        fam = [kv for kv in hits if ", *" in kv[0]]
        for k, v in (fam or hits):
            best = int(v) if rec.get("source_hash") == _lib.source_hash() else "stale"
    return best


def fp64_rates(entry, flops, ms, shape=None, instr_per_pair_step=None, fs=11):
    """The convolution's fp64 figures, labelled for what they are.  `fp64_algorithmic_*`:
    2 (fh fw + LSF taps) flops per voxel -- the arithmetic of the reference's dense stencil
    (lib/run.py:1027-1029), NOT an issue rate: k_conv_rows folds both mirror symmetries of
    the FSF.  `fp64_issued_*`: the fp64 instructions the kernel actually issues per lane
    (csrc/d3d_conv.h: `instr_per_pair_step` per z-pair and march step, halo steps of a strip
    included), against the vector unit's issue peak (78.6 TFLOP/s / 2 flops per FMA)."""
    entry["fp64_algorithmic_tflops"] = round(flops / (ms * 1e-3) / 1e12, 2)
    entry["fp64_algorithmic_frac"] = round(flops / (ms * 1e-3) / 1e12 / FP64_VEC_PEAK_TF, 4)
    if shape is not None and instr_per_pair_step is not None:
        D, H, W = shape
        ngx = (W + 14) // 15                      # launch_conv_rows_t: 15 columns per workgroup,
        ngy = min(max(1, 256 // ngx), H)          # one workgroup per CU, one round
        hy = (H + ngy - 1) // ngy
        steps = (H + hy - 1) // hy * (hy + fs - 1)
        issued = instr_per_pair_step / 2.0 * D * W * steps
        entry["fp64_issued_tinstr_per_s"] = round(issued / (ms * 1e-3) / 1e12, 2)
        entry["fp64_issue_frac"] = round(issued / (ms * 1e-3) / 1e12 / (FP64_VEC_PEAK_TF / 2.0), 4)


def kept_lsf_taps(lsf, bound=1e-16):
    """Taps d3d_set_taps keeps (Engine.set_taps' default): the smallest are dropped within an error
    bound of `bound` * sum|lsf|; taps of EQUAL magnitude (a symmetric LSF's pairs) stay or go
    together, so the cut never makes a symmetric LSF asymmetric."""
    mag = np.sort(np.abs(np.asarray(lsf, dtype=np.float64)))
    limit = bound * mag.sum()
    k = int(np.searchsorted(np.cumsum(mag), limit, side="right"))      # the k smallest fit
    cut = mag[k - 1] if k > 0 else 0.0
    while cut > 0.0 and mag[mag <= cut].sum() > limit:
        below = mag[mag < cut]
        cut = below.max() if below.size else 0.0
    return int(np.count_nonzero(mag > cut))


def traffic_rates(entry, us):
    """traffic_gbs / traffic_frac of a roofline entry whose `traffic` is a byte count."""
    if isinstance(entry.get("traffic"), int) and entry["traffic"] > 0:
        entry["traffic_gbs"] = round(entry["traffic"] / (us * 1e-6) / 1e9, 1)
        entry["traffic_frac"] = round(entry["traffic_gbs"] / HBM_PEAK_GBS, 4)


def beyond_mall_leg(args, local_rank, fs):
    """The MH kernel on a 300x300x256 cube: residual + 1/variance = 369 MB, more
    than the 256 MB Infinity Cache (MALL), so the stream cannot be cache-served as
    the 184 MB working set of the headline cube partly is.  Priced like `roofline`."""
    return mh_cube_leg(args, local_rank, fs, (256, 300, 300),
                       "k_mh_ws, 300x300x256 cube (working set 369 MB > 256 MB MALL)",
                       ("k_mh_ws<256, false, 2, 2, 4,", "true, false, false>"))


def deep_leg(args, local_rank, fs):
    """The sweep on a cube beyond the 512 channels one workgroup takes: 200x200x1024 (655 MB of
    residual + 1/variance) -- k_mh_ws on (window, 256-channel block) workgroups + k_mh_zdecide
    per window, two launches per colour class.  Priced like `roofline`; avg_launch_us is the
    PAIR of launches of one colour class."""
    return mh_cube_leg(args, local_rank, fs, (1024, 200, 200),
                       "k_mh_ws<..., ZBK> + k_mh_zdecide, 200x200x1024 cube (z-blocks of 256 channels)",
                       ("k_mh_ws<256, false, 2, 2, 4,", "true, true, false>"))


def mh_cube_leg(args, local_rank, fs, shape, label, traffic_key):
    from deconv3d_amd import _lib
    D, H, W = shape
    fsf, lsf = build_taps(D, fs)
    steps = max(2, min(args.steps, 10))
    with _lib.Engine((D, H, W), fsf.shape, device=local_rank) as eng:
        eng.set_taps(fsf, lsf)
        data, var, truth, init, min_b, max_b = synthetic_inputs(eng, D, H, W, fsf, 777)
        eng.set_data(data, var, mask=None)
        del data, var
        eng.set_params(init)
        eng.mh_config(min_b, max_b, 0.1, float(max_b[0] ** 2), seed=777, refresh_every=0)
        eng.residual(fetch=False)
        eng.mh_sweeps(2, 1)
        eng.sync()
        eng.timer_start()
        eng.mh_sweeps(steps, 3)
        ms = eng.timer_stop()
        fh, fw = fsf.shape
        ncol = fh * fw
        bytes_per_launch = 3 * 8 * D * window_voxels(H, W, fh, fw) // ncol
        us = ms * 1e3 / (ncol * steps)
        gbs = bytes_per_launch / (us * 1e-6) / 1e9
        return {"kernel": label,
                "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(gbs / HBM_PEAK_GBS, 4),
                # (the committed profile's command runs this leg too: same file, same run)
                "traffic": measured_traffic(traffic_key[0], args.workload, traffic_key[1]),
                "bytes_per_launch": bytes_per_launch, "avg_launch_us": round(us, 2),
                "launches": ncol * steps, "residual_written_every": eng.mh_layers(),
                "value": round(steps * H * W / (ms * 1e-3), 1), "unit_value": "spaxel-updates/s"}


def batched_chains_leg(args, local_rank, chains=16, workload="c2_64x64x64"):
    """BASELINE config 2's cube as an ENSEMBLE on one GPU, through the drop-in API:
    `Run(..., chains=16)` -- 16 independent chains (seeds seed + r) in ONE launch per colour
    class (d3d_mh_sweeps_batch).  A single chain of that cube is a latency chain of 121 small
    launches (49 windows for 1024 workgroup slots); the joint launch carries 16 times the
    windows for about the same latency.  Aggregate rate over the wall time of the runs' device
    calls (`Run.mh_seconds`: the MH loop, setup and outputs excluded); chain r is bit-identical
    to the chain of `Run(..., seed=seed + r)` (tests/test_gpu_run.py)."""
    import logging
    import deconv3d_amd as d3d
    from deconv3d_amd import _lib
    logging.getLogger("deconv3d").setLevel(logging.WARNING)
    D, H, W, fs = WORKLOADS[workload]
    fsf, lsf = build_taps(D, fs)
    with _lib.Engine((D, H, W), fsf.shape, device=local_rank) as eng:
        eng.set_taps(fsf, lsf)
        data, var, truth, init, min_b, max_b = synthetic_inputs(eng, D, H, W, fsf, 4000)
    inst = d3d.Instrument(d3d.VectorLineSpreadFunction(lsf), d3d.ImageFieldSpreadFunction(fsf))
    cube = d3d.MUSE().build_cube(data)
    steps = max(20, min(10 * args.steps, 200))
    kw = dict(variance=var, max_iterations=steps + 1, keep_one_in=steps, min_acceptance_rate=0.,
              seed=4000, device=local_rank, refresh_every=0)
    d3d.Run(cube, inst, chains=chains, **dict(kw, max_iterations=6))          # warm (clocks, allocations)
    many = d3d.Run(cube, inst, chains=chains, **kw)
    one = d3d.Run(cube, inst, **kw)
    rate = chains * steps * H * W / many.mh_seconds
    return {"workload": workload, "chains": chains, "value": round(rate, 1),
            "unit": "spaxel-updates/s", "via": "Run(cube, instrument, chains=%d)" % chains,
            "ms_per_sweep_of_all_chains": round(many.mh_seconds * 1e3 / steps, 4),
            "one_chain_alone": round(steps * H * W / one.mh_seconds, 1),
            "acceptance": round(float(np.mean(many.acceptance_rates)), 3),
            "chain0_equals_the_single_run": bool(np.array_equal(many.chains[0], one.chain)),
            "note": "extra: %d independent chains of one cube in one launch per colour class "
                    "(wall clock of the runs' device calls); not `value`" % chains}


def conv_beyond_mall_leg(args, local_rank, fs):
    """The one-pass convolution on a 600x600x128 cube: 369 MB in + 369 MB out, beyond the
    256 MB Infinity Cache that can hold most of the headline cube's 92 + 92 MB."""
    from deconv3d_amd import _lib
    D, H, W = 128, 600, 600
    fsf, lsf = build_taps(D, fs)
    iters = max(5, min(args.conv_iters, 20))
    rng = np.random.default_rng(5)
    with _lib.Engine((D, H, W), fsf.shape, device=local_rank) as eng:
        eng.set_taps(fsf, lsf)
        eng.upload_slot(_lib.SLOT_DATA, rng.normal(size=(D, H, W)))
        eng.convolve_slots(_lib.SLOT_DATA, _lib.SLOT_SIM)
        eng.sync()
        eng.timer_start()
        for _ in range(iters):
            eng.convolve_slots(_lib.SLOT_DATA, _lib.SLOT_SIM)
        ms = eng.timer_stop() / iters
    nbytes = 2 * 8 * D * H * W
    gbs = nbytes / (ms * 1e-3) / 1e9
    ntaps = kept_lsf_taps(lsf)
    flops = 2.0 * (fsf.size + ntaps) * D * H * W
    out = {"kernel": "k_conv_rows, 600x600x128 cube (369 MB in + 369 MB out > 256 MB MALL)",
           "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(gbs / HBM_PEAK_GBS, 4),
           "traffic": measured_traffic("k_conv_rows<11, 15, true, true, false, 1,", "conv_600x600x128"),
           "algorithmic_bytes": nbytes, "ms_per_conv": round(ms, 4)}
    traffic_rates(out, ms * 1e3)
    fp64_rates(out, flops, ms, (D, H, W), 136, fs)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3_300x300x128", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="ensemble", choices=["ensemble", "tiled"])
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the legs beside the contract line (beyond-MALL cube, uniform "
                         "variance, Gaussian FSF, reference-layout convolution)")
    ap.add_argument("--conv-iters", type=int, default=50)
    ap.add_argument("--no-conv-beyond-mall", action="store_true",
                    help="skip the 600x600x128 convolution leg (tools/profile_round.sh: it runs the "
                         "same kernel as `roofline_conv` and would mix into its counter averages)")
    ap.add_argument("--no-deep", action="store_true",
                    help="skip the 200x200x1024 leg (the z-blocked sweep kernels of cubes beyond 512 channels)")
    ap.add_argument("--tiles", default=None,
                    help="--mode tiled: tile grid TYxTX (default: row strips, N x 1); with "
                         "--gpus 1 the tiles run as contexts of this one process (loop-back)")
    ap.add_argument("--dry-run", action="store_true",
                    help="start the ranks, rendezvous, report rank / seed plumbing and exit "
                         "without touching a GPU (CPU-side test of the launcher path)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for the barrier / MAX-reduce (gloo: rehearsal "
                         "of the multi-process path on a box with fewer GPUs than ranks)")
    ap.add_argument("--no-tiled-leg", action="store_true",
                    help="N > 1, ensemble mode: skip the config-4 leg (one chain tiled over the "
                         "same GPUs, reported as `config4_tiled` beside the ensemble `value`)")
    ap.add_argument("--only-conv-beyond-mall", action="store_true",
                    help="run only the 600x600x128 convolution leg and print its record "
                         "(tools/profile_round.sh: its counters are taken in a run of their own)")
    ap.add_argument("--wait-stdin", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    if args.wait_stdin:
        import torch  # noqa: F401  (paged in while waiting; importing does not touch the GPU)
        if not sys.stdin.readline().strip():
            return                                      # the parent went away: nothing to do
    leg = None
    if world > 1:
        # rank 0's CPU baselines and the extra legs belong to the N = 1 line only
        args.no_cpu = True
        args.no_extras = True
        if (args.mode == "ensemble" and not args.dry_run and not args.no_tiled_leg
                and os.environ.get("D3D_BENCH_TILED_LEG", "1") != "0"):
            try:
                leg = start_tiled_leg(args, rank)
            except OSError as e:                        # the leg is an extra: never fatal
                sys.stderr.write("bench.py: config-4 leg not started: %s\n" % e)

    dist = None
    torch = None
    if world > 1:
        import torch
        import torch.distributed as dist
        ndev = torch.cuda.device_count()
        if args.backend == "nccl" and not args.dry_run:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
            local_rank = local_rank % max(ndev, 1)
            if not args.dry_run:
                torch.cuda.set_device(local_rank)

    if args.dry_run:
        # the plumbing of the self-started job, no GPU: who am I, which chain do I run
        mine = {"rank": rank, "seed": 12345 + rank, "local_rank": local_rank}
        everyone = [mine]
        if dist is not None:
            everyone = [None] * world
            dist.all_gather_object(everyone, mine)
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup,
                              "ranks": [e["rank"] for e in everyone],
                              "seeds": [e["seed"] for e in everyone],
                              "config": {"workload": args.workload,
                                         "parallelism": "1 chain" if world == 1
                                         else "ensemble of %d chains" % world}}))
        return

    if args.mode == "tiled" and world > 1:
        from deconv3d_amd import tiling
        return tiling.bench_tiled(args, rank, local_rank, world, dist, torch)
    if args.mode == "tiled":
        from deconv3d_amd import tiling
        return tiling.bench_tiled_loopback(args, local_rank)

    from deconv3d_amd import _lib

    D, H, W, fs = WORKLOADS[args.workload]
    if args.only_conv_beyond_mall:
        print(json.dumps({"roofline_conv_beyond_mall": conv_beyond_mall_leg(args, local_rank, fs)}))
        return
    fsf, lsf = build_taps(D, fs)
    eng = _lib.Engine((D, H, W), fsf.shape, device=local_rank)
    eng.set_taps(fsf, lsf)
    seed = 12345 + rank                                 # config 5: seeds 12345 + rank
    data, var, truth, init, min_b, max_b = synthetic_inputs(eng, D, H, W, fsf, seed)
    mask = np.ones((H, W))
    eng.set_data(data, var, mask=mask)
    eng.set_params(init)
    ra = float(max_b[0] ** 2)                           # lib/run.py:264-265
    eng.mh_config(min_b, max_b, 0.1, ra, seed=seed, refresh_every=1000)
    err0 = eng.residual() if (rank == 0 and not args.no_cpu) else None
    if err0 is None:
        eng.residual(fetch=False)
    n_spaxels = int(mask.sum())

    def barrier():
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
        eng.sync()

    # ---- warmup, then EXACTLY K timed sweeps ---------------------------------
    sweep = 1
    if args.warmup > 0:
        eng.mh_sweeps(args.warmup, sweep)
        sweep += args.warmup
    barrier()
    t0 = time.perf_counter()
    eng.timer_start()
    accepted = eng.mh_sweeps(args.steps, sweep)
    dev_ms = eng.timer_stop()
    barrier()
    dt = time.perf_counter() - t0
    sweep += args.steps

    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64,
                         device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    total_updates = args.steps * n_spaxels * world
    value = total_updates / dt

    # ---- roofline of the dominant kernel (k_mh, one launch per colour) --------
    fh, fw = fsf.shape
    ncol = sum(1 for c in range(fh * fw) if eng.colour_count(c) > 0)
    bytes_per_sweep = 3 * 8 * D * window_voxels(H, W, fh, fw)     # read err, read 1/var, write err
    launches = ncol * args.steps
    avg_launch_us = dev_ms * 1e3 / launches
    achieved = bytes_per_sweep / ncol / (avg_launch_us * 1e-6) / 1e9
    small = eng.get_option("small_parts") > 0       # launches that do not fill the chip: k_mh_small
    roofline = {"kernel": ("k_mh_small" if small else "k_mh_ws") + " (one launch per colour class)", "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": measured_traffic("k_mh_ws<256, false, 2, 2, 2,", args.workload, "false, false, false>"),
                "bytes_per_launch": bytes_per_sweep // ncol,
                "avg_launch_us": round(avg_launch_us, 2), "launches": launches,
                # `achieved` prices the 24 B per window voxel of SURVEY 8(d) (read residual,
                # read 1/var, write residual); the kernel itself stores the residual only
                # every `residual_written_every` colours and applies the pending updates
                # in registers in between, which is why `traffic` is lower
                "residual_written_every": eng.mh_layers()}
    # HBM bytes the counters saw per launch / the same launch time (the
    # kernel moves fewer bytes than the algorithmic 24 B per window voxel)
    traffic_rates(roofline, avg_launch_us)

    # ---- separable convolution roofline (north_star's second target) ---------
    # cube in -> cube out, device resident.  (a) in the reference's own (D,H,W)
    # layout (d3d_stage_convolve: what d3d_convolve runs between its upload and
    # download), (b) between two spectrum-contiguous slots (the layout of the MH
    # loop, used by the forward model / residual refresh).
    conv_bytes = 2 * 8 * D * H * W                                # cube in -> cube out
    ntaps_lsf = kept_lsf_taps(lsf)
    conv_flops = 2.0 * (fh * fw + ntaps_lsf) * D * H * W

    def conv_entry(ms, kernel, traffic_key, flops=conv_flops, instr=None):
        gbs = conv_bytes / (ms * 1e-3) / 1e9
        e = {"kernel": kernel, "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                "traffic": measured_traffic(traffic_key, args.workload),
                "ms_per_conv": round(ms, 4)}
        fp64_rates(e, flops, ms, (D, H, W) if instr else None, instr, fh)
        return e

    eng.stage_upload(data)
    eng.stage_convolve()                                          # warm
    eng.sync()
    eng.timer_start()
    for _ in range(args.conv_iters):
        eng.stage_convolve()
    stage_ms = eng.timer_stop() / max(args.conv_iters, 1)
    eng.convolve_slots(_lib.SLOT_DATA, _lib.SLOT_SIM)             # warm
    eng.sync()
    eng.timer_start()
    for _ in range(args.conv_iters):
        eng.convolve_slots(_lib.SLOT_DATA, _lib.SLOT_SIM)
    slots_ms = eng.timer_stop() / max(args.conv_iters, 1)
    # (136 / 68 fp64 instructions per z-pair and march step: radial / outer-product FSF + 17-tap
    # LSF epilogue at 128 channels, csrc/d3d_conv.h; other shapes: no issued figure)
    at128 = D == 128 and fh == 11
    roofline_conv = conv_entry(slots_ms, "k_conv_rows (LSF x FSF in one pass, slot layout)",
                               "k_conv_rows<11, 15, true, true, false, 1,",
                               instr=136 if at128 else None)
    roofline_conv["algorithmic_bytes"] = conv_bytes
    roofline_conv_ref_layout = conv_entry(stage_ms, "k_spectral_z + k_spatial_z (reference layout)",
                                          "k_spatial_z")

    # ---- the other two things north_star names, priced with SURVEY 8(d)'s bytes ----
    # forward model (lib/run.py:1011-1029): parameters -> convolved cube = k_lines (raw lines,
    # one exp per voxel) + k_conv_rows (FSF, LSF in its epilogue): H W 3 8 + D H W 8 algorithmic
    # bytes; the 92 MB line cube between the two launches is traffic, not algorithm.
    # chi2 map (lib/run.py:423-424): 0.5 sum_z err^2 / var per spaxel from the CARRIED residual
    # and 1/var (2 D H W 8 + H W 8 bytes; SURVEY's 276.5 MB is the three-cube form data, sim, var).
    eng.forward(fetch=False)
    eng.sync()
    eng.timer_start()
    for _ in range(args.conv_iters):
        eng.forward(fetch=False)
    fwd_ms = eng.timer_stop() / max(args.conv_iters, 1)
    fwd_bytes = H * W * 3 * 8 + D * H * W * 8
    fwd_gbs = fwd_bytes / (fwd_ms * 1e-3) / 1e9
    roofline_forward = {"kernel": "k_lines + k_conv_rows (parameters -> convolved cube, two launches)",
                        "bound": "hbm", "achieved": round(fwd_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(fwd_gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes": fwd_bytes,
                        "ms_per_forward": round(fwd_ms, 4),
                        "ms_line_build": round(fwd_ms - slots_ms, 4)}
    # (both launches' counter bytes: the 92 MB line cube is written by one and read by the other)
    t_lines = measured_traffic("k_lines<256>", args.workload)
    t_conv = measured_traffic("k_conv_rows<11, 15, true, true, false, 1,", args.workload)
    roofline_forward["traffic"] = (t_lines + t_conv) if isinstance(t_lines, int) and isinstance(t_conv, int) \
        else ("stale" if "stale" in (t_lines, t_conv) else None)
    traffic_rates(roofline_forward, fwd_ms * 1e3)
    eng.chi2_map(fetch=False)
    eng.sync()
    eng.timer_start()
    for _ in range(args.conv_iters):
        eng.chi2_map(fetch=False)
    chi_ms = eng.timer_stop() / max(args.conv_iters, 1)
    chi_bytes = 2 * D * H * W * 8 + H * W * 8
    chi_gbs = chi_bytes / (chi_ms * 1e-3) / 1e9
    roofline_chi2 = {"kernel": "k_chi2_map (carried residual, 1/var -> per-spaxel 0.5 chi2)", "bound": "hbm",
                     "achieved": round(chi_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(chi_gbs / HBM_PEAK_GBS, 4), "algorithmic_bytes": chi_bytes,
                     "us_per_map": round(chi_ms * 1e3, 2),
                     "traffic": measured_traffic("k_chi2_map", args.workload)}
    traffic_rates(roofline_chi2, chi_ms * 1e3)

    out = {
        "metric": "spaxel-updates/sec (MH-Gibbs)",
        "value": round(value, 1),
        "unit": "spaxel-updates/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt * 1e3 / args.steps, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": args.workload, "cube": [D, H, W], "fsf": "moffat %dx%d" % (fh, fw)
                   if fs != 9 else "gaussian 9x9",
                   "lsf_taps": ntaps_lsf, "spaxels": n_spaxels,
                   # BASELINE config 3 names MUSELineSpreadFunction: the reference delegates it
                   # to mpdaf (absent here, unpinned), so the taps are the documented stand-in
                   "lsf": "gaussian" if fs == 9 else "analytic MUSE stand-in (mpdaf absent)",
                   "parallelism": "1 chain" if world == 1 else "ensemble of %d chains" % world,
                   "variance": "per-voxel cube (heteroscedastic)"},
        "acceptance": round(accepted / float(args.steps * n_spaxels), 4),
        "roofline": roofline,
        "roofline_conv": roofline_conv,
        "roofline_conv_ref_layout": roofline_conv_ref_layout,
        "roofline_forward": roofline_forward,
        "roofline_chi2": roofline_chi2,
    }

    if rank == 0 and not args.no_extras:
        # PCIe-inclusive: the same sweeps with EVERY sweep streamed to a host chain
        # (keep_one_in = 1: parameters + log ratios, 2.9 MB per sweep) -- device snapshot,
        # copy stream, pinned buffers: the compute stream does not wait for the host
        ks = max(2, min(args.steps, 20))
        chain = np.zeros((sweep + ks + 2, H, W, 3))
        dlog = np.zeros((sweep + ks + 2, H, W))
        eng.mh_sweeps(1, sweep, 1, chain, dlog)      # (allocates the pinned ring, untimed)
        sweep += 1
        eng.sync()
        t1 = time.perf_counter()
        eng.mh_sweeps(ks, sweep, 1, chain, dlog)
        eng.sync()
        out["chain_streaming"] = {
            "keep_one_in": 1, "ms_per_step": round((time.perf_counter() - t1) * 1e3 / ks, 4),
            "bytes_per_step": int(H * W * 4 * 8),
            "note": "same kernel path as `value`, every sweep copied to the host chain"}
        sweep += ks
        del chain, dlog
    if rank == 0:
        out["host"] = host_info()
    if rank == 0 and not args.no_extras and args.workload == "c3_300x300x128":
        out["roofline_beyond_mall"] = beyond_mall_leg(args, local_rank, fs)
        if not args.no_deep:
            out["roofline_deep"] = deep_leg(args, local_rank, fs)
        if not args.no_conv_beyond_mall:
            out["roofline_conv_beyond_mall"] = conv_beyond_mall_leg(args, local_rank, fs)
            # (skipped with it in the profile round: the same kernel names as the headline leg's
            # small-launch variants would mix into the counter averages)
            out["chains_batched"] = batched_chains_leg(args, local_rank)
    if rank == 0 and not args.no_extras:
        # the reference's default variance (Run(variance=None): one constant,
        # lib/run.py:171-178): the MH kernel does not read SLOT_IVAR at all
        assert not eng.variance_is_uniform()
        eng.set_data(data, None, var_scalar=float(np.mean(var)), mask=mask)
        if eng.variance_is_uniform():
            eng.set_params(init)
            eng.residual(fetch=False)
            eng.mh_sweeps(max(args.warmup, 1), 1)
            eng.sync()
            eng.timer_start()
            eng.mh_sweeps(args.steps, 1 + max(args.warmup, 1))
            u_ms = eng.timer_stop()
            u_us = u_ms * 1e3 / launches
            u_gbs = (bytes_per_sweep * 2 // 3) / ncol / (u_us * 1e-6) / 1e9
            out["uniform_variance"] = {
                "value": round(args.steps * n_spaxels / (u_ms * 1e-3), 1),
                "unit": "spaxel-updates/s", "kernel": "k_mh_ws<..., uniform 1/var>",
                "bytes_per_launch": bytes_per_sweep * 2 // 3 // ncol,
                "avg_launch_us": round(u_us, 2), "achieved": round(u_gbs, 1),
                "frac": round(u_gbs / HBM_PEAK_GBS, 4),
                "traffic": measured_traffic("k_mh_ws<256, true, 4, 2, 2,", args.workload, "false, false, false>"),
                "note": "extra: reference default variance=None (one constant); not `value`"}

    if rank == 0 and not args.no_extras:
        # the reference's default FSF (MUSE(): Gaussian, lib/instruments.py:95-107) is an outer
        # product: the spatial pass runs 2*11 instead of 11*11 taps and is bound by HBM
        from deconv3d_amd.spread_functions import gaussian_image
        g = gaussian_image(4.0)            # fwhm 4 px: ceil(6 sigma) = 11 -> 11x11 taps
        if g.shape == (fh, fw):
            eng.set_taps(g, lsf)
            for _ in range(20):                                       # warm (clocks too)
                eng.convolve_slots(_lib.SLOT_DATA, _lib.SLOT_SIM)
            eng.sync()
            eng.timer_start()
            for _ in range(args.conv_iters):
                eng.convolve_slots(_lib.SLOT_DATA, _lib.SLOT_SIM)
            g_ms = eng.timer_stop() / max(args.conv_iters, 1)
            out["roofline_conv_gaussian"] = conv_entry(
                g_ms, "k_conv_rows, outer-product form (Gaussian 11x11 FSF; LSF in the same pass)",
                "k_conv_rows<11, 15, true, true, false, 2,",
                flops=2.0 * (fh + fw + ntaps_lsf) * D * H * W, instr=68 if at128 else None)

    if rank == 0 and not args.no_cpu:
        cores = len(os.sched_getaffinity(0))
        rate, n, secs = cpu_baseline(data, var, mask, fsf, lsf, init, min_b, max_b, err0,
                                     args.cpu_seconds)
        out["cpu_baseline"] = {
            "value": round(rate, 2), "unit": "spaxel-updates/s", "cores": 1, "kind": "port",
            "sample": "%d updates of the same %s workload in %.1f s (oracle.mh_update, "
                      "memory-sane numpy restatement of lib/run.py:369-519; the "
                      "reference-faithful form needs an %.1f TB contributions array and "
                      "cannot run)" % (n, args.workload, secs, H * W * D * H * W * 8 / 1e12),
            "host_cores_visible": cores, "numpy": np.__version__}
        out["vs_cpu"] = round(value / rate, 1)
        ctot, cspec, cspat, n_sp, nz = cpu_baseline_conv(data, fsf, lsf, min(6.0, args.cpu_seconds))
        out["cpu_baseline_conv"] = {
            "value": round(1.0 / ctot, 4), "unit": "cube convolutions/s", "cores": 1, "kind": "port",
            "seconds_per_cube": round(ctot, 3), "spectral_s": round(cspec, 3),
            "spatial_s": round(cspat, 3),
            "sample": "oracle convolve_1d on %d of %d spectra + scipy convolve2d 'same' on %d of %d "
                      "channels of the same cube, scaled to the cube (lib/run.py:999-1031)"
                      % (n_sp, H * W, nz, D),
            "vs_gpu": round(ctot / (slots_ms * 1e-3), 1)}
        frate, fn, fsecs = cpu_baseline_faithful(min(3.0, args.cpu_seconds))
        out["cpu_baseline_faithful"] = {
            "value": round(frate, 2), "unit": "spaxel-updates/s", "cores": 1, "kind": "port",
            "sample": "%d updates in %.1f s of the reference-faithful form (full-cube temporaries + "
                      "contributions array) on config 1, 32x16x16 / 9x9" % (fn, fsecs)}

    eng.close()
    if leg is not None:
        # the ensemble measurement is over and this rank's context is closed: let the
        # children run config 4 on the same GPUs
        try:
            rec = finish_tiled_leg(leg, rank)
        except Exception as e:                          # the leg is an extra: never fatal
            leg[0].kill()
            rec = {"error": "tiled leg: %s: %s" % (type(e).__name__, e)}
        if rank == 0:
            if "value" in rec:
                rec["speedup_vs_one_gpu"] = round(rec["value"] / (value / world), 3)
            out["config4_tiled"] = rec
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
