#!/bin/bash
# Durations of the line-cube kernels by option lines_dense, per cube shape, from rocprofv3's
# kernel trace (one process per shape so that the trace separates them).
#   bash tools/lines_profile.sh > gpurun_out/lines.txt      (on the GPU box)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/lines_prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for sh in 128x300x300 64x300x300 256x300x300 100x300x300 512x200x200; do
    rocprofv3 --kernel-trace --output-format csv -d $OUT/$sh -o t -- python3 $R/tools/lines_time.py $sh > $OUT/$sh.txt 2>&1
    python3 - $OUT/$sh $sh <<'PY'
import collections, csv, sys
import numpy as np
g = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1] + "/t_kernel_trace.csv")):
    n = r["Kernel_Name"]
    if "k_lines" in n or "k_conv_rows" in n or "k_spectral" in n:
        g.setdefault(n.split("(")[0].replace("void d3d::", ""), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
print(sys.argv[2])
for line in open(sys.argv[1] + ".txt"):
    if "lines_dense" in line:
        print("   " + line.strip())
for k, v in g.items():
    print("   %-58s %4d launches, median %6.1f us, min %6.1f" % (k, len(v), np.median(v) / 1e3, min(v) / 1e3))
PY
done
