"""go1ppo_gemm_nt next to torch.mm / addmm (hipBLASLt, TunableOp table of the repo) + the separate ELU kernel it
folds in, on the shapes of the PPO update.  Buffers are rotated so that operands come from HBM, not the 256 MB
Infinity Cache.  Usage (GPU box): python tools/bench_gemm.py [--rows 24576] [--no-tuned]"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "walk-these-ways_amd")
for p in (os.path.join(PKG, "shims"), PKG, REPO):
    sys.path.insert(0, p)
import torch  # noqa: E402
from go1_gym_learn.ppo_cse import fused  # noqa: E402


def timeit(fns, iters=40, warm=4):
    """fns: list of closures over DIFFERENT buffers, called round-robin"""
    for i in range(warm * len(fns)):
        fns[i % len(fns)]()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fns[i % len(fns)]()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3          # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=24576)
    ap.add_argument("--no-tuned", action="store_true")
    args = ap.parse_args()
    if not args.no_tuned:
        from go1_gym_learn.ppo_cse.ppo import _enable_tuned_gemms
        _enable_tuned_gemms()
    M = args.rows
    lib = fused.load_library(os.environ.get("GO1PPO_LIB"))          # (GO1PPO_LIB: a variant build for A/B runs)
    s = torch.cuda.current_stream().cuda_stream
    bf = dict(device="cuda", dtype=torch.bfloat16)
    R = 4                                            # rotating buffer sets
    import ctypes
    print(f"rows = {M}\n{'shape':34s} {'ours us':>9s} {'TF/s':>7s} | {'torch us':>9s} {'TF/s':>7s} {'+elu us':>8s}")
    for name, N, K, lda, elu in (("first layer 2112 -> 1280", 1280, 2112, 2112, (0, 256)), ("adaptation 2112 -> 256", 256, 2112, 2112, True),
                                 ("tail 512 -> 256", 256, 512, 1280, True), ("tail 256 -> 128", 128, 256, 256, True),
                                 ("head 128 -> 64", 64, 128, 128, None), ("tail 256 -> 128 (ld 1280)", 128, 256, 1280, True),
                                 ("dgrad 256 -> 512", 512, 256, 256, "bwd"), ("dgrad 128 -> 256", 256, 128, 128, "bwd")):
        A = [torch.randn(M, lda, **bf)[:, :K] for _ in range(R)]
        B = torch.randn(N, K, **bf) / K ** 0.5
        bias = torch.randn(N, device="cuda")
        C = [torch.zeros(M, N, **bf) for _ in range(R)]
        H = [torch.randn(M, N, **bf) for _ in range(R)]
        if elu == "bwd":
            args_ = [fused.gemm_args(A[i], B, C[i], None, elu_bwd_of=H[i]) for i in range(R)]
        else:
            args_ = [fused.gemm_args(A[i], B, C[i], bias, elu=elu) for i in range(R)]
        import ctypes
        ours = timeit([(lambda g=g: lib.go1ppo_gemm_nt(ctypes.byref(g), s)) for g in args_])
        tt = timeit([(lambda i=i: torch.addmm(bias.to(torch.bfloat16), A[i], B.t(), out=C[i])) for i in range(R)])
        if elu == "bwd":
            te = timeit([(lambda i=i: lib.go1ppo_elu_bwd(C[i].data_ptr(), N, H[i].data_ptr(), N, M, N, None, C[i].data_ptr(), N, s)) for i in range(R)])
        elif elu is not None:
            c1 = N if elu is True else elu[1]
            te = timeit([(lambda i=i: lib.go1ppo_elu_fwd(C[i].data_ptr(), M, c1, N, None, 0, 0, None, 0, 0, s)) for i in range(R)])
        else:
            te = 0.0
        gf = 2 * M * N * K / 1e9
        print(f"{name:34s} {ours:9.1f} {gf / ours:7.0f} | {tt:9.1f} {gf / tt:7.0f} {te:8.1f}")
    print(f"\n{'weight gradient (n x k)':34s} {'tn us':>9s} {'TF/s':>7s} | {'torch us':>9s} {'old us':>8s}")
    import ctypes
    def one(n, k, ld_dz, ld_h, bias=True):
        dz = [torch.randn(M, ld_dz, **bf) for _ in range(R)]
        h = [torch.randn(M, ld_h, **bf) for _ in range(R)]
        out = torch.zeros(n, k, device="cuda")
        outb = torch.zeros(n, k, **bf)
        bg = torch.zeros(n, device="cuda")
        tabs = []
        for i in range(R):
            tab = (fused.WgradProblem * 1)()
            P = tab[0]
            P.dz, P.h, P.dW, P.bias_grad = dz[i].data_ptr(), h[i].data_ptr(), out.data_ptr(), bg.data_ptr() if bias else None
            P.rows, P.ld_dz, P.ld_h, P.n, P.k, P.ldw = M, ld_dz, ld_h, n, k, k
            total = lib.go1ppo_wgrad_tn_plan(tab, 1)
            tabs.append((torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).cuda(), total))
        t_tn = timeit([(lambda t=t: lib.go1ppo_wgrad_tn_batched(t[0].data_ptr(), 1, t[1], s)) for t in tabs])
        t_t = timeit([(lambda i=i: torch.mm(dz[i][:, :n].t(), h[i][:, :k], out=outb)) for i in range(R)])
        t_old = timeit([(lambda i=i: lib.go1ppo_wgrad(dz[i].data_ptr(), ld_dz, h[i].data_ptr(), ld_h, M, n, k, out.data_ptr(), k,
                                                     bg.data_ptr() if bias else None, s)) for i in range(R)]) if n % 64 == 0 and k % 64 == 0 else float("nan")
        gf = 2 * M * n * k / 1e9
        print(f"{n:5d} x {k:5d} (wgs {tabs[0][1]:5d})          {t_tn:9.1f} {gf / t_tn:7.0f} | {t_t:9.1f} {t_old:8.1f}")
    for n, k, ld_dz, ld_h in ((1280, 2112, 1280, 2112), (256, 2112, 256, 2112), (256, 512, 256, 1280), (128, 256, 128, 256),
                              (64, 128, 64, 128), (512, 64, 1280, 64)):
        one(n, k, ld_dz, ld_h)
    # the PPO pass's batched launch: every small weight gradient of one backward pass in one go1ppo_wgrad_tn_batched call
    shapes = [(64, 128, 64, 128), (128, 256, 128, 256), (256, 512, 256, 1280)] * 2 + [(512, 64, 1280, 64), (64, 128, 64, 128), (128, 256, 128, 1280)]
    sets = []
    for _ in range(R):
        tab = (fused.WgradProblem * len(shapes))()
        keep = []
        for P, (n, k, ld_dz, ld_h) in zip(tab, shapes):
            dz, h, out, bg = torch.randn(M, ld_dz, **bf), torch.randn(M, ld_h, **bf), torch.zeros(n, k, device="cuda"), torch.zeros(n, device="cuda")
            keep.append((dz, h, out, bg))
            P.dz, P.h, P.dW, P.bias_grad = dz.data_ptr(), h.data_ptr(), out.data_ptr(), bg.data_ptr()
            P.rows, P.ld_dz, P.ld_h, P.n, P.k, P.ldw = M, ld_dz, ld_h, n, k, k
        total = lib.go1ppo_wgrad_tn_plan(tab, len(shapes))
        sets.append((torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).cuda(), total, keep))
    t_set = timeit([(lambda t=t: lib.go1ppo_wgrad_tn_batched(t[0].data_ptr(), len(shapes), t[1], s)) for t in sets])
    gf = sum(2 * M * n * k for n, k, _, _ in shapes) / 1e9
    print(f"batched PPO-pass set ({len(shapes)} problems, wgs {sets[0][1]}): {t_set:7.1f} us  {gf / t_set:6.0f} TF/s")
    del sets
    print("\nLDS-resident 256 -> 128 -> 64 tails (us per launch)")
    for nets in (1, 2, 3):
        sets = []
        for _ in range(R):
            fw, bw, keep = (fused.Mlp2Fwd * nets)(), (fused.Mlp2Bwd * nets)(), []
            for i in range(nets):
                x = torch.randn(M, 256, **bf); W2 = torch.randn(128, 256, **bf) / 16; b2 = torch.randn(128, **bf); W3 = torch.randn(64, 128, **bf) / 11
                b3 = torch.randn(64, **bf); z2 = torch.zeros(M, 128, **bf); out = torch.zeros(M, 64, **bf); d_out = torch.randn(M, 64, **bf)
                d_z2 = torch.zeros(M, 128, **bf); d_x = torch.zeros(M, 256, **bf)
                P, Q = fw[i], bw[i]
                P.x, P.W2, P.b2, P.W3, P.b3, P.z2, P.out = x.data_ptr(), W2.data_ptr(), b2.data_ptr(), W3.data_ptr(), b3.data_ptr(), z2.data_ptr(), out.data_ptr()
                P.rows, P.ld_x, P.ld_z2, P.ld_out, P.elu_input = M, 256, 128, 64, 1
                Q.d_out, Q.z2, Q.h, Q.W2, Q.W3, Q.d_z2, Q.d_x = d_out.data_ptr(), z2.data_ptr(), x.data_ptr(), W2.data_ptr(), W3.data_ptr(), d_z2.data_ptr(), d_x.data_ptr()
                Q.rows, Q.ld_dout, Q.ld_z2, Q.ld_h, Q.ld_dz2, Q.ld_dx = M, 64, 128, 256, 128, 256
                keep.append((x, W2, b2, W3, b3, z2, out, d_out, d_z2, d_x))
            sets.append((fw, bw, keep))
        tf = timeit([(lambda q=q: lib.go1ppo_mlp2_fwd(q[0], nets, s)) for q in sets])
        tb = timeit([(lambda q=q: lib.go1ppo_mlp2_bwd(q[1], nets, s)) for q in sets])
        print(f"  {nets} net(s): forward {tf:7.1f}   backward {tb:7.1f}")


if __name__ == "__main__":
    main()
