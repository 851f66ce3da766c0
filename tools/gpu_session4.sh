#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
bash tools/profile_round.sh r02 > gpurun_out/s4_profile.log 2>&1
tail -5 gpurun_out/s4_profile.log
python bench.py --mode patch --steps 1 --warmup 1 > gpurun_out/s4_patch.json 2> gpurun_out/s4_patch.err; tail -c 1200 gpurun_out/s4_patch.json; tail -3 gpurun_out/s4_patch.err
python bench.py --steps 5 --warmup 2 --latency > gpurun_out/s4_bench.json 2> gpurun_out/s4_bench.err; head -c 600 gpurun_out/s4_bench.json
