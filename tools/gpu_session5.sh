#!/bin/bash
# GPU session: B = 1 kernel trace (which launches carry the single-image latency), tightened tolerance check.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b1 -o b1 -- python bench.py --batch 1 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/s5_b1.json 2> gpurun_out/s5_b1.err
ls -la gpurun_out/prof_b1 | head
python -m pytest tests -m gpu -q -p no:cacheprovider -s -k "forward_small or forward_sid or alternative or golden_crop or crop" > gpurun_out/s5_pytest.log 2>&1; tail -5 gpurun_out/s5_pytest.log
grep -E "rel|rms|RMS" gpurun_out/s5_pytest.log | head -20
