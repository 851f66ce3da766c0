#!/bin/bash
# Build variants of libucdir_hip.so for same-box A/B runs:  tools/variants.sh name1:"-DFLAG1 -DFLAG2" name2:"" ...
# -> ucdir_amd/variants/libucdir_<name>.so  (git-ignored, travels with gpurun)
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/ucdir_amd/variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -fno-slp-vectorize $flags $R/ucdir_amd/csrc/engine.hip -o $R/ucdir_amd/variants/libucdir_$name.so &
done
wait
ls -la $R/ucdir_amd/variants/
