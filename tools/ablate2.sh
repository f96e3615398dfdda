#!/bin/bash
cd $GRAFT_REPO_ROOT
for a in "" "--zero-actions"; do
  echo -n "actions[$a]: "
  python bench.py --steps 8 --warmup 4 --no-cpu-baseline --sim-only $a 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('launch_ms', round(d['roofline']['launch_ms'],4), 'env-steps/s', round(d['value']))"
done
