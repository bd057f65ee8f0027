#!/bin/bash
# per-kernel durations of the fit objective on the bench shapes (rocprofv3 kernel trace); modes = env settings
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_fit
rm -rf $OUT && mkdir -p $OUT
for shape in "10044 5001 5 3" "381 24963 20 3" "9 24963 20 40"; do
  set -- $shape
  for mode in ${MODES:-default unfused full}; do  # default = fused Gram kernel; unfused = design-matrix kernel + Gram kernel; full = padded full product
    unset PARRM_FIT_FULL_GRAM PARRM_FIT_UNFUSED
    [ $mode = full ] && export PARRM_FIT_FULL_GRAM=1
    [ $mode = unfused ] && export PARRM_FIT_UNFUSED=1
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${1}_$mode -o t -- python3 $GRAFT_REPO_ROOT/scripts/profile_fit.py --periods $1 --n $2 --bw $3 --reps $4 > $OUT/${1}_$mode.log 2>&1
    echo "== P=$1 n=$2 bw=$3 $mode: $(tail -1 $OUT/${1}_$mode.log)"
    python3 - <<PY
import csv, glob
f = glob.glob("$OUT/${1}_$mode/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "fit_" in r["Name"]:
        print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:9.1f}")
PY
  done
done
