#!/bin/bash
# A/B on ONE box: the plane instance of the step kernel against the height-field instance on the same flat terrain, interleaved
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
  for v in plane hf; do
    if [ $v = hf ]; then export GO1_FORCE_HF_INSTANCE=1; else unset GO1_FORCE_HF_INSTANCE; fi
    python bench.py --sim-only --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['roofline']['launch_ms']*1000,1), 'us')"
  done
done
