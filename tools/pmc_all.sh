#!/bin/bash
# PMC pass over the whole headline bench: per kernel (name prefix) the mean of each counter and of the launch duration.
#   bash tools/pmc_all.sh TAG [counter ...]        (GPU box, repo root; output gpurun_out/TAG/pmc_all.txt)
TAG=$1; shift
CTRS=${@:-SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_all
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d /tmp/pmc_all -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 2 --no-cpu-baseline --headline-only > $OUT/pmc_all.log 2>&1
f=$(find /tmp/pmc_all -name "*counter_collection.csv" | head -1)
python - "$f" "$OUT/pmc_all.txt" <<PY
import csv, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(.*", "", r["Kernel_Name"])[:48]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
with open(sys.argv[2], "w") as f:
    for k in sorted(agg, key=lambda k: -sum(dur[k].values())):
        d = dur[k]
        line = f"{k:50s} n={len(d):5d} us={sum(d.values()) / len(d):8.1f} " + " ".join(f"{c}={sum(v) / len(v):.3g}" for c, v in agg[k].items())
        print(line); f.write(line + "\n")
PY
