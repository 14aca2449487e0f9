#!/bin/bash
# Round profile on the GPU box: kernel trace + stats of bench.py, then the two
# HBM-traffic counter passes (FETCH_SIZE and WRITE_SIZE need separate passes:
# MI355X_MICROARCH.md "rocprofv3 PMC slots").  Output under gpurun_out/prof_$1.
set -e
TAG=${1:-r01}
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
CMD="bench.py --steps 5 --warmup 1 --no-cpu --conv-iters 10 --no-conv-beyond-mall"
python $CMD > $OUT/bench_plain.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- python $CMD > $OUT/bench_traced.json 2> $OUT/trace.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT -o fetch -- python $CMD > /dev/null 2> $OUT/fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT -o write -- python $CMD > /dev/null 2> $OUT/write.log
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT -o l2 -- python $CMD > /dev/null 2> $OUT/l2.log || true
python tools/summarize_profile.py $OUT $TAG
cp profiles/${TAG}_summary.md profiles/${TAG}_kernel_stats.csv profiles/${TAG}_traffic.json $OUT/   # (only gpurun_out/ travels back)
