mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_conv_rows.py tests/test_gpu_random_shapes.py -x -q > gpurun_out/r4/s8_pytest.log 2>&1; tail -4 gpurun_out/r4/s8_pytest.log
python tools/conv_shapes.py > gpurun_out/r4/s8_conv_shapes.txt 2>&1; cat gpurun_out/r4/s8_conv_shapes.txt
python bench.py --no-cpu --no-deep --no-conv-beyond-mall > gpurun_out/r4/s8_bench.json 2> gpurun_out/r4/s8_bench.err; python - <<'PY'
import json
rec=json.loads(open('gpurun_out/r4/s8_bench.json').read().strip().splitlines()[-1])
for k in ('roofline_forward','roofline_chi2','roofline_conv','uniform_variance'):
    print(k, rec.get(k))
PY
tail -3 gpurun_out/r4/s8_bench.err
