#!/bin/bash
# GPU session: same-box A/B of two library builds + a parity subset.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "conv_gemm or akgm or statistics or forward_small or forward_sid or batch_is_independent or bit_reproducible or predictor" > gpurun_out/s16_pytest.log 2>&1; tail -3 gpurun_out/s16_pytest.log
bash tools/ab_bench.sh ucdir_amd/libucdir_base.so ucdir_amd/libucdir_hip.so 3 > gpurun_out/s16_ab.log 2>&1; cat gpurun_out/s16_ab.log
