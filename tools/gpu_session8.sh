#!/bin/bash
# GPU session: kernel trace of the full-resolution patch mode (which kernels carry a 1024^2 window).
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_patch -o p -- python bench.py --mode patch --timesteps 4 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/s8_patch.json 2> gpurun_out/s8_patch.err
tail -2 gpurun_out/s8_patch.json | cut -c1-600
