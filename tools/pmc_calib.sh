#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the known-traffic micro-kernels (tools/probes/traffic_calib.hip): separate --pmc passes, GPU box
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_calib
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/traffic_calib $GRAFT_REPO_ROOT/tools/probes/traffic_calib.hip || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/calib_$c -o p -- /tmp/traffic_calib > $OUT/run_$c.log 2>&1
  f=$(find /tmp/calib_$c -name "*counter_collection.csv" | head -1)
  python - "$f" "$c" <<PY | tee -a $OUT/calib.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2]:
        print(sys.argv[2], r["Kernel_Name"].split("(")[0], float(r["Counter_Value"]))
PY
done
