#!/bin/bash
# The data-parallel path over RCCL in a process group of ONE rank (GO1_FORCE_DP=1), all four exchange modes, against the plain run on the
# same box; output gpurun_out/$1/rccl_one_rank.txt.  GPU box, repo root.
OUT=gpurun_out/${1:-rccl}; mkdir -p $OUT
R=$OUT/rccl_one_rank.txt; : > $R
rate() { python - "$1" <<PY
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l); print(f"{d['value']/1e6:.3f} M env-steps/s, {d['ms_per_step']:.2f} ms per iteration")
PY
}
for rep in 1 2; do
python bench.py --steps 20 --warmup 5 --headline-only --no-cpu-baseline > $OUT/plain$rep.log 2>&1; echo "plain single-GPU run #$rep: $(rate $OUT/plain$rep.log)" | tee -a $R
done
i=0
for extra in "" "--grad-dtype bf16" "--zero1" "--grad-dtype bf16 --zero1"; do
  i=$((i+1))
  GO1_FORCE_DP=1 GO1_DP_TRACE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29800+i)) bench.py --gpus 1 --steps 20 --warmup 5 --headline-only --no-cpu-baseline $extra > $OUT/dp$i.log 2>&1
  echo "one-rank RCCL [${extra:-fp32 all-reduce}]: $(rate $OUT/dp$i.log)   $(grep -o '\[dp-trace\].*' $OUT/dp$i.log | head -1)" | tee -a $R
  grep "collectives not capturable" $OUT/dp$i.log | head -1 | tee -a $R
done
GO1_FORCE_DP=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29811 bench.py --gpus 1 --steps 20 --warmup 5 --headline-only --no-cpu-baseline --curriculum-interval 24 > $OUT/dp_k24.log 2>&1
echo "one-rank RCCL [fp32 all-reduce, curriculum interval 24]: $(rate $OUT/dp_k24.log)" | tee -a $R
