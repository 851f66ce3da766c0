#!/bin/bash
# Collect the committed profile evidence of a round on the GPU box:
#   tools/profile_round.sh r01         -> gpurun_out/prof_r01/{trace,pmc_fetch,pmc_write}/ + bench_trace.json
# then (anywhere):  python tools/make_profile_summary.py gpurun_out/prof_r01 r01
# Kernel timing and PMC counters are separate rocprofv3 runs (never --pmc together with a trace domain).
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT/trace $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq1 $OUT/pmc_sq2
cd /tmp && export TMPDIR=/tmp
flat() { find "$1" -name "*.csv" -mindepth 2 -exec mv {} "$1"/ \; ; }
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $R/bench.py --steps 2 --warmup 1 2>/dev/null | tail -1 > $OUT/bench_trace.json
flat $OUT/trace
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $R/bench.py --steps 1 --warmup 0 --timesteps 2 > /dev/null 2>&1
flat $OUT/pmc_fetch
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $R/bench.py --steps 1 --warmup 0 --timesteps 2 > /dev/null 2>&1
flat $OUT/pmc_write
# matrix-core / VALU / LDS counters (SQ block: 8 slots per pass; GRBM separately countable in the same pass)
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE \
  --output-format csv -d $OUT/pmc_sq1 -- python $R/bench.py --steps 1 --warmup 0 --timesteps 2 --no-cpu-baseline > $OUT/pmc_sq1.log 2>&1
flat $OUT/pmc_sq1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU \
  --output-format csv -d $OUT/pmc_sq2 -- python $R/bench.py --steps 1 --warmup 0 --timesteps 2 --no-cpu-baseline > $OUT/pmc_sq2.log 2>&1
flat $OUT/pmc_sq2
tail -2 $OUT/pmc_sq1.log $OUT/pmc_sq2.log
# keep only what the summary needs (the raw kernel trace is tens of MB)
find $OUT -name "*kernel_trace.csv" -delete
ls -la $OUT/*
