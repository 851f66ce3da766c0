#!/bin/bash
# GPU session: register epilogue of conv3x3_halo: parity, then same-box A/B against the previous build.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
python -m pytest tests -m gpu -q -p no:cacheprovider -x -k "conv_gemm or statistics or forward_small or forward_sid or batch_is_independent or bit_reproducible or predictor or alternative" > gpurun_out/s14_pytest.log 2>&1; tail -5 gpurun_out/s14_pytest.log
bash tools/ab_bench.sh ucdir_amd/libucdir_base.so ucdir_amd/libucdir_hip.so 3 > gpurun_out/s14_ab.log 2>&1; cat gpurun_out/s14_ab.log
