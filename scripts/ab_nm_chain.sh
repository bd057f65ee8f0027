#!/bin/bash
# A/B of the Nelder-Mead refinement's two forms inside the bench step (run through gpurun from the repo root):
# device-side chain (PARRM_NM_CHAIN=1) against one parrm_fit_errors_host call per batch (the default), un-profiled
# bench lines first, then the kernel stats of each under rocprofv3.  Output: gpurun_out/ab_nm_chain/
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/ab_nm_chain
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for mode in chain stepped chain stepped; do export PARRM_NM_CHAIN=1;
  if [ $mode = stepped ]; then export PARRM_NM_HOST_STEPPED=1; else unset PARRM_NM_HOST_STEPPED; fi
  python3 "$ROOT/bench.py" --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('$mode', 'ms_per_step', round(d['ms_per_step'],3), 'find_period', round(d['breakdown_ms']['find_period'],3), 'filter_kernel', round(d['breakdown_ms']['filter_kernel'],3))
" | tee -a "$OUT/bench_lines.txt"
done
for mode in chain stepped; do export PARRM_NM_CHAIN=1;
  if [ $mode = stepped ]; then export PARRM_NM_HOST_STEPPED=1; else unset PARRM_NM_HOST_STEPPED; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$mode" -- python3 "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/prof_$mode.log" 2>&1
  f=$(ls "$OUT"/prof_$mode/*/*kernel_stats.csv | head -1)
  cp "$f" "$OUT/kernel_stats_$mode.csv"
  head -25 "$f" | cut -c1-160
done
