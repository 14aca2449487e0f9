#!/bin/bash
# Round profile on the GPU box: kernel trace + stats of bench.py, then the two
# HBM-traffic counter passes (FETCH_SIZE and WRITE_SIZE need separate passes:
# MI355X_MICROARCH.md "rocprofv3 PMC slots").  Output under gpurun_out/prof_$1.
# Three commands: the default bench line (headline cube and its extra legs), the 600x600x128
# convolution alone (its kernel names are the headline leg's: own run, own traffic file), and
# BASELINE config 2's cube (k_mh_small: kernel trace only).
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
# ---- the convolution beyond the Infinity Cache, alone
O2=$OUT/conv600
mkdir -p $O2
C2="bench.py --only-conv-beyond-mall --conv-iters 10"
python $C2 > $O2/bench_plain.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O2 -o trace -- python $C2 > $O2/bench_traced.json 2> $O2/trace.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O2 -o fetch -- python $C2 > /dev/null 2> $O2/fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O2 -o write -- python $C2 > /dev/null 2> $O2/write.log
python tools/summarize_profile.py $O2 $TAG conv_600x600x128 _conv600 "python $C2"
# ---- config 2's cube: the small colour launches (k_mh_small), kernel trace + stats
O3=$OUT/c2
mkdir -p $O3
C3="bench.py --workload c2_64x64x64 --steps 20 --warmup 2 --no-cpu --no-extras --conv-iters 5"
python $C3 > $O3/bench_plain.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O3 -o trace -- python $C3 > $O3/bench_traced.json 2> $O3/trace.log
python tools/summarize_profile.py $O3 $TAG c2_64x64x64 _c2 "python $C3"
cp profiles/${TAG}_* $OUT/   # (only gpurun_out/ travels back)
