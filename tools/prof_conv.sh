#!/bin/bash
# FETCH/WRITE/L2 counters of the convolution kernels alone (tools/conv_only.py).
export TMPDIR=/tmp
OUT=gpurun_out/prof_conv_${1:-x}
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python tools/conv_only.py 20 > $OUT/plain.log 2> $OUT/trace.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch -- python tools/conv_only.py 20 > /dev/null 2> $OUT/fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o write -- python tools/conv_only.py 20 > /dev/null 2> $OUT/write.log
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT -o l2 -- python tools/conv_only.py 20 > /dev/null 2> $OUT/l2.log || true
python - <<PY
import csv, collections, glob, os
out="$OUT"
def agg(pat, counter):
    d=collections.defaultdict(list)
    for f in glob.glob(os.path.join(out,"**",pat), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"]==counter: d[r["Kernel_Name"].split("(")[0][:60]].append(float(r["Counter_Value"]))
    return d
f=agg("fetch_counter_collection.csv","FETCH_SIZE"); w=agg("write_counter_collection.csv","WRITE_SIZE")
h=agg("l2_counter_collection.csv","TCC_HIT_sum"); m=agg("l2_counter_collection.csv","TCC_MISS_sum")
for k in f:
    fb=2*1024*sum(f[k])/len(f[k])/1e6; wb=1024*sum(w[k])/len(w[k])/1e6 if k in w else 0
    hr = sum(h[k])/(sum(h[k])+sum(m[k])) if k in h and k in m and sum(h[k])+sum(m[k])>0 else -1
    print("%-60s fetch %.1f MB write %.1f MB  L2 hit %.3f  (n=%d)"%(k,fb,wb,hr,len(f[k])))
for fcsv in glob.glob(os.path.join(out,"**","trace_kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(fcsv)):
        print("%-60s calls %s avg %.2f us min %.2f max %.2f"%(r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
PY
