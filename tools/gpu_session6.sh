#!/bin/bash
# GPU session: split-K for under-filled conv grids: parity, B = 1 trace, same-box A/B at B = 16.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "conv_gemm or akgm or statistics or forward_small or forward_sid or batch_is_independent or alternative" > gpurun_out/s6_pytest.log 2>&1; tail -5 gpurun_out/s6_pytest.log
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_b1s -o b1 -- python bench.py --batch 1 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/s6_b1.json 2> gpurun_out/s6_b1.err
for i in 1 2; do
  for v in "UCDIR_SPLITK=0" "UCDIR_SPLITK=1" "UCDIR_SPLITK_WGS=256"; do
    echo "$v $(env $v python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'],2), ' '.join('%s:%.2f'%(k['kernel'][:14],k['ms']) for k in d['roofline']['all_kernels'][:8]))")"
  done
done > gpurun_out/s6_ab.log 2>&1
cat gpurun_out/s6_ab.log
for v in "UCDIR_SPLITK=0" "UCDIR_SPLITK=1"; do
  env $v python bench.py --steps 2 --warmup 1 --no-cpu-baseline --latency 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d.get('latency'))"
done > gpurun_out/s6_lat.log 2>&1
cat gpurun_out/s6_lat.log
