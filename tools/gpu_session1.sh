#!/bin/bash
# GPU session: per-phase cycle stamps of the hot kernels, same-box A/B of two builds, detailed bench.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
T=$R/ucdir_amd/libucdir_hip_timing.so
{
for a in "akgm 16 288 288 64 2" "akgm 16 144 144 128 2" "akgm 16 72 72 256 2" "akgm 16 36 36 512 2" \
         "conv 16 288 288 64 64 2" "conv 16 288 288 192 64 2" "conv 16 144 144 128 128 2" "conv 16 144 144 384 128 2" \
         "conv 16 72 72 512 256 2" "conv 16 36 36 1024 512 2"; do
  echo "== $a"; UCDIR_LIB=$T timeout 300 python tools/bench_op.py $a 2>&1 | grep -E "TIMING|done" | tail -4
done
} > gpurun_out/s1_timing.log 2>&1
bash tools/ab_bench.sh ucdir_amd/libucdir_base.so ucdir_amd/libucdir_hip.so 2 > gpurun_out/s1_ab.log 2>&1
UCDIR_PROF_DETAIL=1 python bench.py --steps 3 --warmup 1 --latency > gpurun_out/s1_bench.json 2> gpurun_out/s1_bench.err
tail -3 gpurun_out/s1_ab.log; head -c 1500 gpurun_out/s1_bench.json
python -m pytest tests -m gpu -q -p no:cacheprovider -s -k "attention or jpeg or real_patch_window or forward_small or forward_sid" > gpurun_out/s1_pytest.log 2>&1; tail -5 gpurun_out/s1_pytest.log
