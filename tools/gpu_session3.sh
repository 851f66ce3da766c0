#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
for L in ucdir_amd/libucdir_v2.so ucdir_amd/libucdir_hip.so; do
  echo "=== tests with $L"
  UCDIR_LIB=$R/$L python -m pytest tests/test_hip_gpu.py -m gpu -q -p no:cacheprovider -k "akgm or forward_small or forward_sid or batch_is or bit_repro or alternative" 2>&1 | tail -4
done > gpurun_out/s3_pytest.log 2>&1
cat gpurun_out/s3_pytest.log
T=$R/ucdir_amd/libucdir_hip_timing.so
{
for a in "akgm 16 288 288 64 2" "akgm 16 144 144 128 2" "akgm 16 72 72 256 2"; do
  echo "== $a"; UCDIR_LIB=$T timeout 300 python tools/bench_op.py $a 2>&1 | grep -E "TIMING|done" | tail -3
done
} > gpurun_out/s3_timing.log 2>&1
for i in 1 2; do
for L in ucdir_amd/libucdir_base.so ucdir_amd/libucdir_v2.so ucdir_amd/libucdir_hip.so; do
    UCDIR_LIB=$PWD/$L python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$L', round(d['value'],2), ' '.join('%s:%.2f'%(k['kernel'][:14],k['ms']) for k in d['roofline']['all_kernels'][:8]))"
done; done > gpurun_out/s3_ab.log 2>&1
cat gpurun_out/s3_ab.log
