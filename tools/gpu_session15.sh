#!/bin/bash
# GPU session: cost-model split-K choice: parity of the conv tests, B = 1 trace and latency, B = 16 sanity.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "conv_gemm or statistics or forward_small or forward_sid or batch_is_independent" > gpurun_out/s15_pytest.log 2>&1; tail -3 gpurun_out/s15_pytest.log
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --latency 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('latency'))" > gpurun_out/s15_lat.log 2>&1
cat gpurun_out/s15_lat.log
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b1u -o b1 -- python bench.py --batch 1 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/s15_b1.json 2> gpurun_out/s15_b1.err
