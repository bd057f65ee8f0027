#!/bin/bash
# Second-tier SQ counters of the filter kernel (clock, instruction fetch, LDS queueing).
set -e -o pipefail
TAG=${1:-sqx}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "GRBM_GUI_ACTIVE GRBM_COUNT SQ_WAIT_IFETCH SQ_IFETCH SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_CYCLES" \
           "SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_LDS_MEM_VIOLATIONS SQ_LDS_ATOMIC_RETURN SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VALU_MFMA_BUSY_CYCLES SQ_ACCUM_PREV" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$OUT/g$i" -- python3 "$ROOT/scripts/profile_filter.py" --kernel phase --reps 1 > "$OUT/g$i.log" 2>&1 || { echo "group $i failed"; tail -5 "$OUT/g$i.log"; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
for f in glob.glob(out + "/g*/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
with open(out + "/summary.txt", "w") as fh:
    for k, d in tot.items():
        if "filter" not in k: continue
        fh.write(k + " durations(ms) " + str(dur[k]) + "\n")
        for c, v in sorted(d.items()): fh.write(f"  {c} {v:.6g}\n")
print(open(out + "/summary.txt").read())
PY
