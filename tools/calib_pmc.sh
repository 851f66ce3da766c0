#!/bin/bash
# WRITE_SIZE / FETCH_SIZE of a launch that writes exactly 1 GiB and reads nothing (tools/calib_pmc.py): two --pmc passes.
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/calib; rm -rf $OUT; mkdir -p $OUT/w $OUT/f
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/w -- python $R/tools/calib_pmc.py > $OUT/w.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python $R/tools/calib_pmc.py > $OUT/f.log 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
for tag, cn in (("w", "WRITE_SIZE"), ("f", "FETCH_SIZE")):
    v = []
    for f in glob.glob(sys.argv[1] + "/" + tag + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "fill_normal" in r["Kernel_Name"] and r["Counter_Name"] == cn: v.append(float(r["Counter_Value"]))
    if v: print("%s of fill_normal_kernel (writes 1 GiB = 1048576 KiB, reads nothing): %.0f per launch over %d launches -> ratio %.4f" % (cn, sum(v) / len(v), len(v), sum(v) / len(v) / 1048576.0))
PY
