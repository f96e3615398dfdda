"""Probe: which operand layout lets hipBLASLt / rocBLAS run the first-layer weight gradient dW(1280 x 2112) = dY^T X
(24576 rows) fastest?  TunableOp tunes every variant; buffers rotate so operands come from HBM.  GPU box only."""
import os
import sys
import torch
from torch.cuda import tunable

tunable.enable(True)
tunable.tuning_enable(True)
tunable.set_filename("/tmp/wgrad_layouts.csv", insert_device_ordinal=False)
tunable.set_max_tuning_duration(60)
tunable.set_rotating_buffer_size(512)

M, N, K = 24576, 1280, 2112
bf = dict(device="cuda", dtype=torch.bfloat16)
R = 4
dY = [torch.randn(M, N, **bf) for _ in range(R)]
X = [torch.randn(M, K, **bf) for _ in range(R)]
dYt = [d.t().contiguous() for d in dY]
Xt = [x.t().contiguous() for x in X]
out_nk = torch.zeros(N, K, **bf)
out_kn = torch.zeros(K, N, **bf)
out32 = torch.zeros(N, K, device="cuda")


def timeit(fns, iters=40, warm=8):
    for i in range(warm):
        fns[i % len(fns)]()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fns[i % len(fns)]()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


gf = 2 * M * N * K / 1e9
cases = {
    "dY^T X          (row-strided both, out n x k)": [lambda i=i: torch.mm(dY[i].t(), X[i], out=out_nk) for i in range(R)],
    "X^T dY          (row-strided both, out k x n)": [lambda i=i: torch.mm(X[i].t(), dY[i], out=out_kn) for i in range(R)],
    "dYt Xt^T        (K-contiguous both, out n x k)": [lambda i=i: torch.mm(dYt[i], Xt[i].t(), out=out_nk) for i in range(R)],
    "Xt dYt^T        (K-contiguous both, out k x n)": [lambda i=i: torch.mm(Xt[i], dYt[i].t(), out=out_kn) for i in range(R)],
    "dYt X           (A K-contiguous, B strided)": [lambda i=i: torch.mm(dYt[i], X[i], out=out_nk) for i in range(R)],
    "dY^T Xt^T       (A strided, B K-contiguous)": [lambda i=i: torch.mm(dY[i].t(), Xt[i].t(), out=out_nk) for i in range(R)],
}
for name, fns in cases.items():
    t = timeit(fns)
    print(f"{name:50s} {t:8.1f} us  {gf / t:7.0f} TF/s", flush=True)
t = timeit([lambda i=i: torch.transpose_copy(dY[i], 0, 1, out=dYt[i]) for i in range(R)])
print(f"transpose dY (24576 x 1280 bf16)                   {t:8.1f} us")
for r in tunable.get_results():
    print(r)
