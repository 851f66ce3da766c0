#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
python -m pytest tests/test_hip_gpu.py -m gpu -q -p no:cacheprovider -x -k "conv_gemm or akgm or forward_small or forward_sid or statistics or batch_is or alternative" > gpurun_out/s2_pytest.log 2>&1; tail -4 gpurun_out/s2_pytest.log
T=$R/ucdir_amd/libucdir_hip_timing.so
{
for a in "akgm 16 288 288 64 2" "akgm 16 144 144 128 2" "akgm 16 72 72 256 2"; do
  echo "== $a"; UCDIR_LIB=$T timeout 300 python tools/bench_op.py $a 2>&1 | grep -E "TIMING|done" | tail -3
done
} > gpurun_out/s2_timing.log 2>&1
bash tools/ab_bench.sh ucdir_amd/libucdir_base.so ucdir_amd/libucdir_hip.so 2 > gpurun_out/s2_ab.log 2>&1
cat gpurun_out/s2_ab.log
