#!/bin/bash
# A/B of step-kernel build variants on the GPU box: every walk-these-ways_amd/csrc/variants/*.so is put in place of
# libgo1sim.so in turn and timed with the sim-only diagnostic bench (and optionally the phase profile).
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/variants}
mkdir -p "$OUT"
C=walk-these-ways_amd/csrc
cp $C/libgo1sim.so /tmp/libgo1sim.keep
for v in $C/variants/*.so; do
  n=$(basename $v .so)
  cp $v $C/libgo1sim.so
  timeout 100 python bench.py --sim-only --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline > $OUT/$n.log 2>&1
  echo "$n $(tail -1 $OUT/$n.log | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]/1e6,3), "M env-steps/s")' 2>&1 | tail -1)"
  if [ -n "$PHASES" ]; then timeout 150 python tools/phase_profile.py --steps 32 > $OUT/$n.phases.txt 2>&1; grep -E "cycles per| 1 | 2 | 4 | 5 |18 " $OUT/$n.phases.txt; fi
done
cp /tmp/libgo1sim.keep $C/libgo1sim.so
