#!/bin/bash
# Per-kernel durations of one command under rocprofv3 (kernel trace + stats only):  tools/kprof.sh <tag> <command...>
# prints the top lines of the stats table; raw CSVs under gpurun_out/kprof_<tag>/
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; shift
OUT=$R/gpurun_out/kprof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- "$@" > $OUT/cmd.log 2>&1
find $OUT -name "*.csv" -mindepth 2 -exec mv {} $OUT/ \;
find $OUT -name "*kernel_trace.csv" -delete
echo "== $TAG"; tail -2 $OUT/cmd.log
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*kernel_stats.csv")
if not f: print("no stats csv"); sys.exit()
rows = list(csv.DictReader(open(f[0])))
for r in rows[:12]:
    print("%-60s calls %5s avg %10.1f us  %5.1f %%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
