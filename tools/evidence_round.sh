#!/bin/bash
# Committed evidence of a round with the final build (run through gpurun):  tools/evidence_round.sh [tag]
# kernel trace + PMC passes (tools/profile_round.sh), patch mode, headline + latency; then, anywhere:
#   python tools/make_profile_summary.py gpurun_out/prof_<tag> <tag>; cp gpurun_out/ev_patch.json profiles/<tag>_patch_1gpu.json
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
TAG=${1:-r02}
bash tools/profile_round.sh $TAG > gpurun_out/ev_profile.log 2>&1
tail -3 gpurun_out/ev_profile.log
python bench.py --mode patch --steps 1 --warmup 1 > gpurun_out/ev_patch.json 2> gpurun_out/ev_patch.err; tail -c 300 gpurun_out/ev_patch.json
python bench.py --steps 5 --warmup 2 --latency > gpurun_out/ev_bench.json 2> gpurun_out/ev_bench.err; head -c 400 gpurun_out/ev_bench.json
