#!/bin/bash
# Gram-kernel + reduce durations of one K = 41 optimiser batch (9 candidates) under forced sample splits
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_split
rm -rf $OUT && mkdir -p $OUT
for shape in "9 24963 20 40" "10 10001 10 40" "12 5001 5 40"; do
  set -- $shape
  for ns in ${SPLITS:-0 8 16 24 32 48 64}; do
    unset PARRM_FIT_X_NSPLIT
    [ $ns != 0 ] && export PARRM_FIT_X_NSPLIT=$ns
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${1}_$ns -o t -- python3 $GRAFT_REPO_ROOT/scripts/profile_fit.py --periods $1 --n $2 --bw $3 --reps $4 > $OUT/${1}_$ns.log 2>&1
    echo "== P=$1 n=$2 bw=$3 nsplit=$ns (0 = planned)"
    python3 - <<PY
import csv, glob
f = glob.glob("$OUT/${1}_$ns/**/*kernel_stats.csv", recursive=True)[0]
tot = 0.0
for r in csv.DictReader(open(f)):
    if "fit_" in r["Name"]:
        tot += float(r['AverageNs'])/1e3
        print(f"  {r['Name'][:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:9.1f}")
print(f"  sum of averages {tot:.1f} us")
PY
  done
done
