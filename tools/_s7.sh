mkdir -p gpurun_out/r4
python -m pytest tests -m gpu -x -q > gpurun_out/r4/s7_pytest.log 2>&1; tail -5 gpurun_out/r4/s7_pytest.log
python tools/tile_rank_time.py > gpurun_out/r4/s7_tile_rank.txt 2>&1; cat gpurun_out/r4/s7_tile_rank.txt
python examples/quickstart.py 2000 8 > gpurun_out/r4/s7_quickstart.txt 2>&1; tail -4 gpurun_out/r4/s7_quickstart.txt
python bench.py > gpurun_out/r4/s7_bench.json 2> gpurun_out/r4/s7_bench.err; tail -c 3000 gpurun_out/r4/s7_bench.json; tail -3 gpurun_out/r4/s7_bench.err
