#!/bin/bash
# Same-box A/B of two builds of libucdir_hip.so:  tools/ab_bench.sh <libA> <libB> [rounds]
# Alternates bench.py runs (box-to-box variance is larger than most kernel changes).
A=$1; B=$2; R=${3:-2}
for i in $(seq 1 $R); do
  for L in "$A" "$B"; do
    UCDIR_LIB=$PWD/$L python bench.py --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$L', round(d['value'],2), ' '.join('%s:%.2f'%(k['kernel'][:14],k['ms']) for k in d['roofline']['all_kernels'][:7]))"
  done
done
