# The non-headline workloads of bench.py, one line each (profiles/r03_bench_other_workloads.txt).  Run on the GPU box.
for w in c1_32x16x16 c2_64x64x64 d64_300x300x64 d256_300x300x256 d512_300x300x512 d1024_300x300x1024; do
  python bench.py --workload $w --no-cpu --no-extras --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; c=d.get('roofline_conv',{})
print('%-20s %10.0f updates/s  %8.4f ms per sweep  %7.2f us per launch  roofline.frac %.3f | conv %.1f us  frac %.3f' % (d['config']['workload'], d['value'], d['ms_per_step'], r['avg_launch_us'], r['frac'], c.get('ms_per_conv',0)*1e3, c.get('frac',0)))"
done
