#!/bin/bash
# PMC passes for go1_step_kernel (sim-only bench): instruction mix / stalls, then HBM fetch and write sizes (separate passes)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$1
mkdir -p $OUT
run() {
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$name -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --sim-only $PMC_EXTRA > /tmp/pmc_$name.log 2>&1
  f=$(find /tmp/pmc_$name -name "*counter_collection.csv" | head -1)
  python - "$f" "$OUT/$name.txt" <<PY
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kernel_Name"].startswith("go1_step_kernel")]
agg = collections.defaultdict(list)
for r in rows: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[2], "w") as f:
    for k, v in agg.items():
        line = "%s: mean per launch %.1f over %d launches" % (k, sum(v) / len(v), len(v))
        print(line); f.write(line + "\n")
PY
}
run insts SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run stalls SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU
run fetch FETCH_SIZE
run write WRITE_SIZE
