#!/bin/bash
# GPU session: committed round-2 evidence with the final build: kernel trace + PMC passes, patch mode, headline + latency.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
bash tools/profile_round.sh r02 > gpurun_out/s12_profile.log 2>&1
tail -3 gpurun_out/s12_profile.log
python bench.py --mode patch --steps 1 --warmup 1 > gpurun_out/s12_patch.json 2> gpurun_out/s12_patch.err; tail -c 300 gpurun_out/s12_patch.json
python bench.py --steps 5 --warmup 2 --latency > gpurun_out/s12_bench.json 2> gpurun_out/s12_bench.err; head -c 400 gpurun_out/s12_bench.json
