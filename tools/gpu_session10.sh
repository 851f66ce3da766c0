#!/bin/bash
# GPU session: res_conv as tail workgroups of conv1's launch: parity + same-box A/B.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "forward_small or forward_sid or batch_is_independent or alternative or bit_reproducible" > gpurun_out/s10_pytest.log 2>&1; tail -5 gpurun_out/s10_pytest.log
for i in 1 2 3; do
  for v in "UCDIR_NO_TAIL_RES=1" "UCDIR_X=1"; do
    echo "$v $(env $v python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'],2), ' '.join('%s:%.2f'%(k['kernel'][:14],k['ms']) for k in d['roofline']['all_kernels'][:8]))")"
  done
done > gpurun_out/s10_ab.log 2>&1
cat gpurun_out/s10_ab.log
