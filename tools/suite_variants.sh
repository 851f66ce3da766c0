#!/bin/bash
# GPU session: the GPU suite under forced launch-configuration variants (every conv grid split / no tail workgroups / no flash).
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
for v in "UCDIR_SPLITK_WGS=1000000" "UCDIR_NO_TAIL_RES=1" "UCDIR_NO_FLASH=1"; do
  echo "== $v"
  env $v python -m pytest tests -m gpu -q -p no:cacheprovider -k "not alternative and not fp16 and not jpeg and not real_patch_window and not rescale" 2>&1 | tail -6
done > gpurun_out/variants.log 2>&1
cat gpurun_out/variants.log
