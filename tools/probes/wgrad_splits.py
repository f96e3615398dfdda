"""Probe: the first-layer weight gradients as hipBLASLt batched GEMMs over row chunks (manual split-K) for other row-block sizes:
the adaptation pass's dW (256 x 2112), and the PPO pass split into the critic's rows (512) and the adaptation + actor rows (768)
so that the critic's part can run beside the actor's backward chain.  Partials summed by go1ppo_sum_partials.  GPU box only."""
import os
import sys
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P_ = os.path.join(R_, "walk-these-ways_amd")
for p in (os.path.join(P_, "shims"), P_, R_):
    sys.path.insert(0, p)
import torch
from torch.cuda import tunable
from go1_gym_learn.ppo_cse import fused

tunable.enable(True)
tunable.tuning_enable(True)
tunable.set_filename("/tmp/wgrad_splits.csv", insert_device_ordinal=False)
tunable.set_max_tuning_duration(40)
tunable.set_rotating_buffer_size(512)
lib = fused.load_library()
s = torch.cuda.current_stream().cuda_stream
M, K = 24576, 2112
bf = dict(device="cuda", dtype=torch.bfloat16)
R = 4
dYfull = [torch.randn(M, 1280, **bf) for _ in range(R)]
X = [torch.randn(M, K, **bf) for _ in range(R)]


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


for name, c0, c1, splits in (("adaptation rows (256)", 0, 256, (4, 8, 12, 16, 24, 32)), ("critic rows (512)", 768, 1280, (2, 4, 6, 8)),
                             ("adaptation + actor rows (768)", 0, 768, (2, 4, 6, 8)), ("all rows (1280)", 0, 1280, (4,))):
    N = c1 - c0
    gf = 2 * M * N * K / 1e9
    g32 = torch.zeros(N, K, device="cuda")
    for b in splits:
        out = torch.zeros(b, N, K, **bf)
        f = [lambda i=i: torch.bmm(dYfull[i][:, c0:c1].view(b, M // b, N).transpose(1, 2), X[i].view(b, M // b, K), out=out) for i in range(R)]
        t = timeit(f)
        ts = timeit([lambda: lib.go1ppo_sum_partials(out.data_ptr(), b, out.stride(0), N, K, g32.data_ptr(), 0, 0, 0, s)])
        print(f"{name:32s} b {b:2d}: bmm {t:7.1f} us ({gf / t:5.0f} TF/s) + sum {ts:5.1f} us = {t + ts:7.1f} us", flush=True)
for r in tunable.get_results():
    print(r)
