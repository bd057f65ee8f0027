#!/bin/bash
# kernel-trace of find_period alone (run on the GPU box through gpurun)
set -e -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_fp
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$ROOT/scripts/profile_find_period.py" > "$OUT/stdout.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(r["Name"][:60].ljust(60), r["Calls"].rjust(5), f'{float(r["TotalDurationNs"])/1e6:9.2f} ms', f'{float(r["AverageNs"])/1e3:9.1f} us', r["Percentage"])
PY
