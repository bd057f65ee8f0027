#!/bin/bash
# A/B two builds of the library on one box: alternate processes, same workload.
# usage: ab_libs.sh OTHER.so [dtype] [other-first|tree-first]
other=$1; dt=${2:-f64}; order=${3:-other-first}
run_other() { python scripts/ab_filter.py --shapes 4,2 --rounds 5 --dtype $dt --lib $other 2>&1 | tail -1 | sed 's/^/other: /'; }
run_tree() { python scripts/ab_filter.py --shapes 4,2 --rounds 5 --dtype $dt 2>&1 | tail -1 | sed 's/^/tree:  /'; }
for i in 1 2 3; do
  if [ "$order" = other-first ]; then run_other; run_tree; else run_tree; run_other; fi
done
