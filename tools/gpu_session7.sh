#!/bin/bash
# GPU session: full GPU suite + latency legs + B = 1 trace after the small-grid changes.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/s7_pytest.log 2>&1; tail -8 gpurun_out/s7_pytest.log
python bench.py --steps 2 --warmup 1 --no-cpu-baseline --latency 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d.get('latency'))" > gpurun_out/s7_lat.log 2>&1
cat gpurun_out/s7_lat.log
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b1t -o b1 -- python bench.py --batch 1 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/s7_b1.json 2> gpurun_out/s7_b1.err
