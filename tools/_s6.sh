mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_parity.py -x -q -k "small_launch" > gpurun_out/r4/s6_pytest.log 2>&1; tail -3 gpurun_out/r4/s6_pytest.log
EXP=$PWD/deconv3d_amd/csrc/exp/libdeconv3d_hip.so
for spec in "64x64x64" "32x16x16,9" "128x37x300 parts=1" "128x150x150 parts=1" "128x40x40"; do
  for inf in 4 8 0; do
    python tools/mh_tail.py $spec mh_inflight=$inf >> gpurun_out/r4/s6_tail_default.log 2>&1
    DECONV3D_HIP_LIB=$EXP python tools/mh_tail.py $spec mh_inflight=$inf >> gpurun_out/r4/s6_tail_exp.log 2>&1
  done
done
cat gpurun_out/r4/s6_tail_default.log
