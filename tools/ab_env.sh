#!/bin/bash
# Same-box A/B of environment switches:  tools/ab_env.sh [rounds] "VAR=1 ..." "VAR2=1" ...   ("-" = default environment)
R=${1:-2}; shift
for i in $(seq 1 $R); do
  for E in "$@"; do
    if [ "$E" = "-" ]; then EE=""; else EE="$E"; fi
    env $EE python bench.py --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[%s]' % '$E', round(d['value'],2), ' '.join('%s:%.2f'%(k['kernel'][:14],k['ms']) for k in d['roofline']['all_kernels'][:8]))"
  done
done
