#!/bin/bash
# kernel-phase ablation (diagnostic): rebuild libgo1sim.so with one phase compiled out, time the sim-only bench
cd $GRAFT_REPO_ROOT/walk-these-ways_amd/csrc
for flag in NONE GO1_ABLATE_POST GO1_ABLATE_TORQUE GO1_ABLATE_PHYSICS; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-hip-fp32-correctly-rounded-divide-sqrt -D$flag -o libgo1sim.so go1sim.hip 2>/dev/null
  cd $GRAFT_REPO_ROOT
  echo -n "$flag: "
  python bench.py --steps 5 --warmup 2 --no-cpu-baseline --sim-only 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('launch_ms', round(d['roofline']['launch_ms'],4), 'env-steps/s', round(d['value']))"
  cd $GRAFT_REPO_ROOT/walk-these-ways_amd/csrc
done
