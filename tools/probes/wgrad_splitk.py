"""Probe: first-layer weight gradient dW(1280 x 2112) = dY^T X (24576 rows) as a BATCHED GEMM over row chunks (manual
split-K: `b` partial products, summed afterwards) — does hipBLASLt fill the chip better that way?  GPU box only."""
import torch
from torch.cuda import tunable

tunable.enable(True)
tunable.tuning_enable(True)
tunable.set_filename("/tmp/wgrad_splitk.csv", insert_device_ordinal=False)
tunable.set_max_tuning_duration(60)
tunable.set_rotating_buffer_size(512)

M, N, K = 24576, 1280, 2112
bf = dict(device="cuda", dtype=torch.bfloat16)
R = 4
dY = [torch.randn(M, N, **bf) for _ in range(R)]
X = [torch.randn(M, K, **bf) for _ in range(R)]
g32 = torch.zeros(N, K, device="cuda")


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
ref = torch.mm(dY[0].t().float(), X[0].float())
for b in (1, 2, 3, 4, 6, 8, 12, 16):
    out = torch.zeros(b, N, K, **bf)
    out2 = torch.zeros(b, K, N, **bf)
    f1 = [lambda i=i: torch.bmm(dY[i].view(b, M // b, N).transpose(1, 2), X[i].view(b, M // b, K), out=out) for i in range(R)]
    f2 = [lambda i=i: torch.bmm(X[i].view(b, M // b, K).transpose(1, 2), dY[i].view(b, M // b, N), out=out2) for i in range(R)]
    t1, t2 = timeit(f1), timeit(f2)
    f1[0]()
    ts = timeit([lambda: torch.sum(out, dim=0, dtype=torch.float32, out=g32)])
    err = (g32 - ref).abs().max().item() / ref.abs().max().item()
    print(f"batch {b:2d}: dY^T X {t1:7.1f} us ({gf / t1:5.0f} TF/s)   X^T dY {t2:7.1f} us ({gf / t2:5.0f} TF/s)   sum of partials -> fp32 {ts:6.1f} us   rel err {err:.2e}", flush=True)
for r in tunable.get_results():
    print(r)
