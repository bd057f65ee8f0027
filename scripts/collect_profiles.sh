#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root: collects the rocprofv3 evidence that
# profiles/ keeps.  Usage: bash scripts/collect_profiles.sh <tag>
#   1. kernel-trace + stats of the default bench command
#   2. HBM traffic of the filter kernel: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes
#      (they do not fit one pass: MI355X_MICROARCH.md, rocprofv3 PMC slots)
set -e -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench" -- python3 "$ROOT/bench.py" --steps ${STEPS:-3} --warmup ${WARMUP:-1} --no-cpu-baseline > "$OUT/bench_stdout.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/scripts/profile_filter.py" --reps 1 > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/scripts/profile_filter.py" --reps 1 > "$OUT/pmc_write.log" 2>&1
echo "profiles collected under $OUT"
