"""Probe: would the first-layer weight gradient dW1 = dY1^T X run faster on hipBLASLt with K-contiguous operands (dY1^T (1280 x M) and
X^T (2112 x M): the forward GEMM's "Alik_Bljk" layout, 1.39 PFLOP/s in situ) than with the M-major operands it has now ("Ailk_Bjlk",
0.93 PFLOP/s)?  X^T could be made once per update (the mini-batch blocks are gathered once), dY1^T would cost a transposed store or a
transpose pass per mini-batch.  Batched over row chunks (manual split-K) and as one GEMM.  GPU box only."""
import os
import sys
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P_ = os.path.join(R_, "walk-these-ways_amd")
for p in (os.path.join(P_, "shims"), P_, R_):
    sys.path.insert(0, p)
import torch
from torch.cuda import tunable

tunable.enable(True)
tunable.tuning_enable(True)
tunable.set_filename("/tmp/wgrad_layouts.csv", insert_device_ordinal=False)
tunable.set_max_tuning_duration(60)
tunable.set_rotating_buffer_size(512)
M, K, N = 24576, 2112, 1280
bf = dict(device="cuda", dtype=torch.bfloat16)
R = 3
dY = [torch.randn(M, N, **bf) for _ in range(R)]
X = [torch.randn(M, K, **bf) for _ in range(R)]
dYT = [d.t().contiguous() for d in dY]


def timeit(fns, iters=30, warm=6):
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


gf = 2 * M * N * K / 1e3        # -> TFLOP/s with microseconds
for b in (1, 2, 4, 6, 8):
    m = M // b
    out = torch.zeros(b, N, K, **bf)
    cur = [lambda i=i: torch.bmm(dY[i].view(b, m, N).transpose(1, 2), X[i].view(b, m, K), out=out) for i in range(R)]
    # K-contiguous, chunk-major: dYT_b (b, N, m) and XT_b (b, K, m), both contiguous (what a transposing producer would write)
    dYTb = [d.view(b, m, N).transpose(1, 2).contiguous() for d in dY]
    XTb = [x.view(b, m, K).transpose(1, 2).contiguous() for x in X]
    nt = [lambda i=i: torch.bmm(dYTb[i], XTb[i].transpose(1, 2), out=out) for i in range(R)]
    t0, t1 = timeit(cur), timeit(nt)
    ref = torch.bmm(dY[0].view(b, m, N).transpose(1, 2).float(), X[0].view(b, m, K).float()).sum(0)
    nt[0]()
    torch.cuda.synchronize()
    err = float((out.float().sum(0) - ref).abs().max() / ref.abs().max())
    print(f"chunks {b}: M-major operands {t0:7.1f} us ({gf / t0:5.0f} TF/s) | K-contiguous operands {t1:7.1f} us ({gf / t1:5.0f} TF/s)   [check {err:.1e}]", flush=True)
    del dYTb, XTb
tr = timeit([lambda i=i: dYT[i].copy_(dY[i].t()) for i in range(R)])
print(f"transpose pass of dY1 (torch copy_ of a (24576 x 1280) bf16 view): {tr:.1f} us")
for r in tunable.get_results():
    print(r)
