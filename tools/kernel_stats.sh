#!/bin/bash
# Static resource usage of the simulator kernels (no GPU needed): VGPRs, SGPRs, spills, scratch, LDS, code size.
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/go1sim_isa}
mkdir -p "$OUT" && cd "$OUT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize ${EXTRA_FLAGS} -c --cuda-device-only -save-temps=obj \
  -o go1sim.o "$REPO/walk-these-ways_amd/csrc/go1sim.hip" 2>&1 | grep -v warning | head -5
S=go1sim-hip-amdgcn-amd-amdhsa-gfx950.s
for k in go1_step_kernel go1_aux_kernel; do
  echo "== $k"
  grep -B8 -A14 "name:           $k" $S | grep -E "vgpr_count|sgpr_count|spill_count|private_segment_fixed_size|group_segment_fixed_size" | sed 's/^ */  /'
  awk "/^$k:/,/s_endpgm/" $S | grep -cE "^\s+(v_|s_|ds_|global_|scratch_|buffer_)" | sed 's/^/  static instructions: /'
done
