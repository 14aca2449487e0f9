"""Long chain at BASELINE config 3 on one GPU: acceptance, reduced chi2 and the
drift of the carried residual against a from-scratch one (lib/run.py:521-534).
Writes a small report (profiles/<tag>_soak.txt when run through gpurun)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, build_taps, synthetic_inputs  # noqa: E402
from deconv3d_amd import _lib  # noqa: E402

n_sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
chunk = 250 if n_sweeps <= 5000 else 1000
D, H, W, fs = WORKLOADS["c3_300x300x128"]
fsf, lsf = build_taps(D, fs)
eng = _lib.Engine((D, H, W), fsf.shape)
eng.set_taps(fsf, lsf)
data, var, truth, init, min_b, max_b = synthetic_inputs(eng, D, H, W, fsf, 12345)
eng.set_data(data, var)
eng.set_params(init)
eng.mh_config(min_b, max_b, [0., 0.5, 0.2], float(max_b[0] ** 2), seed=12345, refresh_every=1000)
eng.residual(fetch=False)
print("soak: %d sweeps of %dx%dx%d, jump amplitudes (0, 0.5, 0.2), refresh every 1000" % (n_sweeps, D, H, W))
print("%8s %10s %12s %14s %10s" % ("sweep", "accept", "red.chi2", "drift(max|d|)", "M upd/s"))
s = 1
snaps = []                                     # parameter maps at the end of every chunk
t_all = time.perf_counter()
while s <= n_sweeps:
    n = min(chunk, n_sweeps - s + 1)
    t0 = time.perf_counter()
    acc = eng.mh_sweeps(n, s)
    dt = time.perf_counter() - t0
    s += n
    carried = eng.download_slot(_lib.SLOT_ERR)
    _, total = eng.chi2_map()
    saved = eng.get_params()
    snaps.append(saved)
    fresh = eng.residual()                      # from scratch (also what the refresh does)
    drift = float(np.max(np.abs(carried - fresh)))
    print("%8d %10.4f %12.4f %14.3e %10.2f" % (s - 1, acc / float(n * H * W), 2 * total / data.size,
                                               drift, n * H * W / dt / 1e6), flush=True)
p = eng.get_params()
live = truth[..., 0] > 2.0
print("median |c - c_true| over bright spaxels: %.3f channels" % np.median(np.abs(p[..., 1] - truth[..., 1])[live]))
print("median |w - w_true| over bright spaxels: %.3f channels" % np.median(np.abs(p[..., 2] - truth[..., 2])[live]))
# the reference's estimator is the MEAN of the last 20 % of the chain (extract_parameters,
# lib/run.py:581-593), not the last state: the last state scatters by one posterior sigma
tail = np.array(snaps[int(0.8 * len(snaps)):])
pm, ps = tail.mean(0), tail.std(0)
print("last-20%% mean of %d snapshots: median |c - c_true| %.3f, |w - w_true| %.3f channels; "
      "scatter of the snapshots (posterior sigma) median c %.3f, w %.3f"
      % (len(tail), np.median(np.abs(pm[..., 1] - truth[..., 1])[live]),
         np.median(np.abs(pm[..., 2] - truth[..., 2])[live]),
         np.median(ps[..., 1][live]), np.median(ps[..., 2][live])))
# what the data constrain is the CONVOLVED model: last state and tail mean against the truth's
sim_truth = eng.simulate(truth, convolved=True)
for name, pp in (("last state", p), ("tail mean of parameters", pm)):
    d = eng.simulate(pp, convolved=True) - sim_truth
    print("convolved model of the %s vs truth: rms %.4f, max %.4f (noise sigma %.4f, peak %.3f)"
          % (name, np.sqrt(np.mean(d ** 2)), np.abs(d).max(), np.sqrt(np.median(var)), sim_truth.max()))
print("total wall %.1f s" % (time.perf_counter() - t_all))
eng.close()
