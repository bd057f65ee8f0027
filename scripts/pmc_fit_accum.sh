#!/bin/bash
# HBM traffic of the Gram kernel on the stage-3 / stage-1 shapes: FETCH_SIZE and WRITE_SIZE in separate --pmc passes
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_fit
rm -rf $OUT && mkdir -p $OUT
for shape in "381 24963 20" "10044 5001 5"; do
  set -- $shape
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/${1}_$ctr -o t -- python3 $GRAFT_REPO_ROOT/scripts/profile_fit.py --periods $1 --n $2 --bw $3 --reps 2 > $OUT/${1}_$ctr.log 2>&1
    python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/${1}_$ctr/**/*counter_collection.csv", recursive=True)[0]
tot = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "$ctr" and "fit_" in r["Kernel_Name"]:
        tot[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k, v in tot.items():
    print(f"P=$1 $ctr {k:60s} launches {len(v)}  per launch {sum(v)/len(v)/1e6:9.3f} GB (raw kB/1e6; FETCH_SIZE x2 on gfx950 per the guide)")
PY
  done
done
