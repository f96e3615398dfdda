#!/bin/bash
# Build step-kernel variants that differ only in compiler flags into walk-these-ways_amd/csrc/variants/ (for
# tools/variant_bench.sh on the GPU box).  Each variant = the flags of __graft_entry__.build() plus its own.
#   tools/build_variants.sh            # the stock set below
#   tools/build_variants.sh name "-flag1 -flag2" ...
cd "$(dirname "$0")/.."
C=walk-these-ways_amd/csrc
mkdir -p $C/variants
BASE="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-slp-vectorize -Iinclude"
# every variant carries the hash of the sources + its own flags (go1sim_version(): a stale variant in an A/B run shows)
stamp() { python3 -c "import sys; sys.path.insert(0, \".\"); import __graft_entry__ as g; print(g.source_hash(g.sim_sources(), g.SIM_FLAGS + sys.argv[1].split()))" "$1"; }
build() {
  /opt/rocm/bin/hipcc $BASE $2 -DGO1_SOURCE_HASH="\"$(stamp "$2")\"" -o $C/variants/$1.so $C/go1sim.hip && echo "built $1 ($2)"
}
if [ $# -ge 2 ]; then
  while [ $# -ge 2 ]; do build "$1" "$2"; shift 2; done
else
  build base ""
  build ftz "-fgpu-flush-denormals-to-zero"                  # fp32 divisions lose their frexp scaling (-2 % static VALU)
  build unroll0 "-mllvm -unroll-threshold=0"
  build sched_ilp "-mllvm -amdgpu-schedule-metric-bias=100"
fi
