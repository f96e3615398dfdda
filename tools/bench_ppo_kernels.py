"""Micro-benchmark of the fused PPO kernels on the real shapes (M = 24576) next to the torch / hipBLASLt op each one
replaces.  Usage (GPU box): python tools/bench_ppo_kernels.py [--rows 24576]"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "walk-these-ways_amd")
for p in (os.path.join(PKG, "shims"), PKG, REPO):
    sys.path.insert(0, p)
import torch  # noqa: E402
from go1_gym_learn.ppo_cse import fused  # noqa: E402


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3          # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=24576)
    args = ap.parse_args()
    M = args.rows
    lib = fused.load_library()
    s = torch.cuda.current_stream().cuda_stream
    bf = dict(device="cuda", dtype=torch.bfloat16)
    print(f"rows = {M}")
    print("wgrad (n x k)      fused us   torch.mm us   GF    fused TF/s")
    for n, k, ld_dz, ld_h in ((256, 512, 256, 1280), (128, 256, 128, 256), (64, 128, 64, 128), (512, 64, 1280, 64), (128, 256, 128, 1280)):
        dz = torch.randn(M, ld_dz, **bf)
        h = torch.randn(M, ld_h, **bf)
        out = torch.zeros(n, k, device="cuda")
        outb = torch.zeros(n, k, **bf)
        bias = torch.zeros(n, device="cuda")
        tf = timeit(lambda: lib.go1ppo_wgrad(dz.data_ptr(), ld_dz, h.data_ptr(), ld_h, M, n, k, out.data_ptr(), k, bias.data_ptr(), s))
        tt = timeit(lambda: torch.mm(dz[:, :n].t(), h[:, :k], out=outb))
        gf = 2 * M * n * k / 1e9
        print(f"{n:4d} x {k:4d}       {tf:8.1f}   {tt:10.1f}   {gf:5.1f}   {gf / tf * 1e3 / 1e3:8.1f}")
    print("elementwise            us     GB moved   TB/s")
    for cols, ld in ((1024, 1280), (512, 512), (256, 256), (128, 128), (64, 64)):
        y = torch.randn(M, ld, **bf)
        d = torch.randn(M, ld, **bf)
        bias = torch.zeros(cols, device="cuda")
        t1 = timeit(lambda: lib.go1ppo_elu_fwd(y.data_ptr(), M, cols, ld, None, 0, 0, None, 0, 0, s))
        t2 = timeit(lambda: lib.go1ppo_elu_bwd(d.data_ptr(), ld, y.data_ptr(), ld, M, cols, None, d.data_ptr(), ld, s))
        t3 = timeit(lambda: torch.nn.functional.elu(y[:, :cols]))
        gb1, gb2 = 2 * M * cols * 2 / 1e9, 3 * M * cols * 2 / 1e9
        print(f"elu_fwd {cols:4d}: {t1:7.1f} us {gb1 / t1 * 1e3:6.2f} TB/s | elu_bwd+colsum: {t2:7.1f} us {gb2 / t2 * 1e3:6.2f} TB/s | torch elu (out of place): {t3:7.1f} us")
    print("GEMMs (hipBLASLt through torch)")
    x = torch.randn(M, 2104, **bf)
    for name, a, b in (("W1 fwd  (M x 2104) @ (2104 x 1280)", x, torch.randn(1280, 2104, **bf).t()),
                       ("tail    (M x 512) @ (512 x 256)", torch.randn(M, 512, **bf), torch.randn(256, 512, **bf).t()),
                       ("tail    (M x 256) @ (256 x 128)", torch.randn(M, 256, **bf), torch.randn(128, 256, **bf).t()),
                       ("head    (M x 128) @ (128 x 64)", torch.randn(M, 128, **bf), torch.randn(64, 128, **bf).t()),
                       ("dgrad   (M x 256) @ (256 x 512)", torch.randn(M, 256, **bf), torch.randn(256, 512, **bf)),
                       ("dgrad   (M x 64) @ (64 x 128)", torch.randn(M, 64, **bf), torch.randn(64, 128, **bf))):
        out = torch.zeros(a.shape[0], b.shape[1], **bf)
        t = timeit(lambda: torch.mm(a, b, out=out))
        gf = 2 * a.shape[0] * a.shape[1] * b.shape[1] / 1e9
        print(f"{name:40s} {t:8.1f} us  {gf / t * 1e3 / 1e3:7.1f} TF/s")
    dY = torch.randn(M, 1280, **bf)
    tmp = torch.zeros(1280, 2104, **bf)
    t = timeit(lambda: torch.mm(dY.t(), x, out=tmp))
    print(f"{'W1 wgrad (1280 x M) @ (M x 2104) bf16 out':40s} {t:8.1f} us  {2 * M * 1280 * 2104 / 1e9 / t * 1e3 / 1e3:7.1f} TF/s")
    idx = torch.randperm(4 * M, device="cuda")[:M]
    hist = torch.randn(4 * M, 2104, **bf)
    t = timeit(lambda: torch.index_select(hist, 0, idx, out=x))
    print(f"gather X: {t:.1f} us ({2 * M * 2104 * 2 / 1e9 / t * 1e3:.2f} TB/s)")


if __name__ == "__main__":
    main()
