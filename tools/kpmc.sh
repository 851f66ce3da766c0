#!/bin/bash
# SQ / LDS counters of one command, two rocprofv3 --pmc passes (no trace domains):  tools/kpmc.sh <tag> <kernel-substring> <command...>
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=$1; KSUB=$2; shift; shift
OUT=$R/gpurun_out/kpmc_$TAG
rm -rf $OUT; mkdir -p $OUT/p1 $OUT/p2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -- "$@" > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d $OUT/p2 -- "$@" > $OUT/p2.log 2>&1
python3 - "$OUT" "$KSUB" <<'PY'
import csv, glob, sys, collections
out, ksub = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if ksub in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
if not tot: print("no rows for", ksub); sys.exit()
per = {k: tot[k] / n[k] for k in tot}
for k in sorted(per): print("%-28s %14.0f" % (k, per[k]))
g = per.get("GRBM_GUI_ACTIVE", 0)
if g:
    print("MFMA busy            %.3f" % (per["SQ_VALU_MFMA_BUSY_CYCLES"] / (g / 8 * 1024 * 8) * 8 if False else per["SQ_VALU_MFMA_BUSY_CYCLES"] / (g * 1024 / 8)))
    print("waves waiting        %.3f (WAIT_ANY / WAVE_CYCLES)  issue-stalled %.3f" % (per["SQ_WAIT_ANY"] / per["SQ_WAVE_CYCLES"], per["SQ_WAIT_INST_ANY"] / per["SQ_WAVE_CYCLES"]))
if "SQ_INSTS_MFMA" in per:
    print("VALU per MFMA        %.2f   LDS conflict / active %.3f" % (per["SQ_INSTS_VALU"] / max(per["SQ_INSTS_MFMA"], 1), per["SQ_LDS_BANK_CONFLICT"] / max(per["SQ_LDS_IDX_ACTIVE"], 1)))
PY
