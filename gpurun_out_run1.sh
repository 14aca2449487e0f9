set -e
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
for cfg in "1024 0" "512 0" "256 0" "1024 8" "512 16" "256 32"; do
  set -- $cfg
  echo "== NT=$1 MAXIT=$2"
  D3D_MH_NT=$1 D3D_MH_MAXIT=$2 timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu 2>&1 | tail -1 | tee -a gpurun_out/variants.log
done
