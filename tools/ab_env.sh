#!/bin/bash
# GPU session: environment-variable A/B on one build:  tools/ab_env.sh "VAR=a" "VAR=b" ...
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
for i in 1 2 3; do
  for v in "$@"; do
    echo "$v $(env $v python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'],2), ' '.join('%s:%.2f'%(k['kernel'][:14],k['ms']) for k in d['roofline']['all_kernels'][:9]))")"
  done
done > gpurun_out/abenv_ab.log 2>&1
cat gpurun_out/abenv_ab.log
