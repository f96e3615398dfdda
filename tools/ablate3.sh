#!/bin/bash
# finer ablation of go1_step_kernel on the GPU box: rebuild csrc/libgo1sim.so with -D switches / flags, time sim-only
cd $GRAFT_REPO_ROOT/walk-these-ways_amd/csrc
run() {
  name=$1; shift
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-hip-fp32-correctly-rounded-divide-sqrt "$@" -o libgo1sim.so go1sim.hip 2>/dev/null || { echo "$name: build failed"; return; }
  for a in "" "--zero-actions"; do
    echo -n "$name actions[$a]: "
    (cd $GRAFT_REPO_ROOT && python bench.py --steps 8 --warmup 4 --no-cpu-baseline --sim-only $a 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('launch_ms', round(d['roofline']['launch_ms'],4))")
  done
}
run baseline
run fastdiv -fno-hip-fp32-correctly-rounded-divide-sqrt
run no_pgs -DGO1_ABLATE_PGS
run no_pgs_no_delassus -DGO1_ABLATE_PGS -DGO1_ABLATE_DELASSUS
run no_cand -DGO1_ABLATE_CAND
run no_torque -DGO1_ABLATE_TORQUE
run no_post -DGO1_ABLATE_POST
run baseline_again
