#!/bin/bash
# GPU session: full-resolution patch mode, 3 vs 6 windows per engine batch; headline sanity after the bench.py refactor.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
for pb in 0 6; do
  python bench.py --mode patch --steps 1 --warmup 1 --patch-batch $pb > gpurun_out/s9_patch_$pb.json 2> gpurun_out/s9_patch_$pb.err
  python -c "
import json
d=json.loads(open('gpurun_out/s9_patch_$pb.json').read().strip().splitlines()[-1])
r=d['roofline']
print('$pb', d['value'], d['ms_per_step'], d['config']['workspace_bytes_rank0']/1e9, r['kernel'], round(r['achieved']), r['avg_launch_us'])
print(' '.join('%s:%.1f/%d'%(k['kernel'][:16],k['ms'],k['launches']) for k in r['all_kernels'][:9]))"
done
python bench.py --steps 2 --warmup 1 --no-cpu-baseline | cut -c1-400
