set -e
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for hy in 11 16 22 33; do
D3D_MARCH_HY=$hy timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('HY=$hy', d['roofline_conv']['ms_per_conv'], d['roofline_conv_slots']['ms_per_conv'])"
done
