#!/bin/bash
# Offline look at the generated comb kernel: precompile (hipRTC), then hipcc -save-temps on the dumped source.
# usage: scripts/comb_isa.sh [fs] [f_art]   -> /tmp/comb/*.s
set -e
cd "$(dirname "$0")/.."
rm -rf /tmp/comb && mkdir -p /tmp/comb
python scripts/comb_precompile.py ${1:-22000} ${2:-130} /tmp/comb
cd /tmp/comb
F=$(ls comb_*.hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -include hip/hip_runtime.h -c $F -o c.o -save-temps \
    -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "GPRs|Scratch|error|Occupancy|LDS" || true
S=${F%.hip}-hip-amdgcn-amd-amdhsa-gfx950.s
cp $S kernel.s
for pat in v_readfirstlane ds_read_b64 ds_read2 ds_write v_add_f64 v_fma_f64 "s_waitcnt vmcnt" buffer_load_dwordx4 buffer_store s_barrier scratch_ v_mov_b32; do
    printf "%-24s %s\n" "$pat" "$(grep -c -- "$pat" kernel.s)"
done
