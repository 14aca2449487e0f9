set -e
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
for d in 0 1; do
D3D_MH_DEFER=$d timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline'])"
done
