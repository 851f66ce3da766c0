#!/bin/bash
# GPU session: full GPU suite, then same-box A/B of the small-grid splits at B = 16.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/s11_pytest.log 2>&1; tail -4 gpurun_out/s11_pytest.log
for i in 1 2; do
  for v in "UCDIR_SPLITK=0" "UCDIR_X=1"; do
    echo "$v $(env $v python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value'],2), ' '.join('%s:%.2f'%(k['kernel'][:14],k['ms']) for k in d['roofline']['all_kernels'][:8]))")"
  done
done > gpurun_out/s11_ab.log 2>&1
cat gpurun_out/s11_ab.log
