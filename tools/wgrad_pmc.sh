#!/bin/bash
# PMC passes for wgrad_tn_batched_kernel on the adaptation first-layer problem (256 x 2112, 24576 rows); GPU box, repo root.
cat > /tmp/wg_driver.py <<PY
import os, sys, ctypes
R = os.environ["GRAFT_REPO_ROOT"]; P = os.path.join(R, "walk-these-ways_amd")
for p in (os.path.join(P, "shims"), P, R): sys.path.insert(0, p)
import torch
from go1_gym_learn.ppo_cse import fused
lib = fused.load_library(); s = torch.cuda.current_stream().cuda_stream
M, n, k = 24576, int(sys.argv[1]), 2112
bf = dict(device="cuda", dtype=torch.bfloat16)
dz = [torch.randn(M, n, **bf) for _ in range(4)]; h = [torch.randn(M, k, **bf) for _ in range(4)]
out = torch.zeros(n, k, device="cuda")
tabs = []
for i in range(4):
    tab = (fused.WgradProblem * 1)(); Pb = tab[0]
    Pb.dz, Pb.h, Pb.dW, Pb.bias_grad = dz[i].data_ptr(), h[i].data_ptr(), out.data_ptr(), None
    Pb.rows, Pb.ld_dz, Pb.ld_h, Pb.n, Pb.k, Pb.ldw = M, n, k, n, k, k
    total = lib.go1ppo_wgrad_tn_plan(tab, 1)
    tabs.append((torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).cuda(), total))
for it in range(12):
    t = tabs[it % 4]; lib.go1ppo_wgrad_tn_batched(t[0].data_ptr(), 1, t[1], s)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/wgpmc_$1
mkdir -p $OUT
run() {
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/wgpmc_$name -o p -- python /tmp/wg_driver.py $N > /tmp/wgpmc_$name.log 2>&1
  f=$(find /tmp/wgpmc_$name -name "*counter_collection.csv" | head -1)
  python - "$f" "$OUT/$name.txt" <<PY
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "wgrad_tn" in r["Kernel_Name"]]
agg = collections.defaultdict(list)
for r in rows: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(sys.argv[2], "w") as f:
    for k, v in agg.items():
        line = "%s: mean per launch %.1f over %d launches" % (k, sum(v) / len(v), len(v))
        print(line); f.write(line + "\n")
PY
}
N=${2:-256}
run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
run b SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD
