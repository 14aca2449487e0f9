"""The kernels that are NOT bound by HBM, run a few times each for a
counter pass of rocprofv3:  k_lines + k_conv_rows (forward model), k_spectral_z + k_spatial_z
(reference-layout convolution), k_conv_rows alone (slot convolution), at 300x300x128.

    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY \\
              SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d OUT -o c -- python3 tools/issue_counters.py
    python3 tools/issue_counters.py --summarize OUT
"""
import collections
import csv
import glob
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def summarize(out):
    f = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void d3d::", "")
        if not any(s in k for s in ("k_lines", "k_conv_rows", "k_spatial_z", "k_spectral_z", "k_chi2_map")):
            continue
        acc.setdefault(k, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        m = {n: np.median(v) for n, v in c.items()}
        wc = m.get("SQ_WAVE_CYCLES", float("nan"))
        print("%s  (%d dispatches)" % (k, len(next(iter(c.values())))))
        print("    wave cycles %.3g (quad-cycles); of them issuing any instruction %.2f, vector ALU %.2f, parked on "
              "s_waitcnt / barrier %.2f, issue-stalled %.2f" % (
                  wc, m.get("SQ_ACTIVE_INST_ANY", 0) / wc, m.get("SQ_ACTIVE_INST_VALU", 0) / wc,
                  m.get("SQ_WAIT_ANY", 0) / wc, m.get("SQ_WAIT_INST_ANY", 0) / wc))
        print("    vector instructions %.4g; GRBM_GUI_ACTIVE / 8 = %.0f cycles" % (
            m.get("SQ_INSTS_VALU", 0), m.get("GRBM_GUI_ACTIVE", 0) / 8))


if len(sys.argv) > 2 and sys.argv[1] == "--summarize":
    summarize(sys.argv[2])
    sys.exit(0)

import bench as B  # noqa: E402
from deconv3d_amd import _lib  # noqa: E402

D, H, W = 128, 300, 300
fsf, lsf = B.build_taps(D, 11)
with _lib.Engine((D, H, W), fsf.shape) as eng:
    eng.set_taps(fsf, lsf)
    data, var, truth, init, min_b, max_b = B.synthetic_inputs(eng, D, H, W, fsf, 12345)
    eng.set_data(data, var)
    eng.set_params(init)
    eng.stage_upload(data)
    for _ in range(6):
        eng.forward(fetch=False)
        eng.stage_convolve()
        eng.residual(fetch=False)
        eng.chi2_map(fetch=False)
    eng.sync()
