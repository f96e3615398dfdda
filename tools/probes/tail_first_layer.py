"""Probe: how fast is "row tile resident in LDS (ELU applied on the way in), weights streamed from L2 straight into MFMA operands" for
the actor's and critic's 512 -> 256 layer at 24576 rows?  go1ppo_tail_fwd restricted to ONE layer per net, with and without its
(2-byte, fragment-shaped) global stores, 32 and 64 rows per workgroup (GO1PPO_TAIL_BM64=1).  Against: go1ppo_elu_fwd + go1ppo_gemm_nt_pair
(the update's current path).  GPU box only."""
import ctypes
import os
import sys
R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P_ = os.path.join(R_, "walk-these-ways_amd")
for p in (os.path.join(P_, "shims"), P_, R_):
    sys.path.insert(0, p)
import torch
from go1_gym_learn.ppo_cse import fused

lib = fused.load_library(os.environ.get("GO1PPO_LIB"))
s = torch.cuda.current_stream().cuda_stream
M = int(os.environ.get("ROWS", "24576"))
bf = dict(device="cuda", dtype=torch.bfloat16)
R = 3


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


Y1 = [torch.randn(M, 1280, **bf) for _ in range(R)]
W = [torch.randn(256, 512, **bf) / 22 for _ in range(2)]
b = [torch.randn(256, **bf) for _ in range(2)]
lat = torch.randn(M, 64, **bf)
wz = torch.randn(512, 64, **bf)
for store in (False, True):
    sets = []
    for i in range(R):
        a = fused.TailArgs()
        a.num_nets = 2
        outs = [torch.zeros(M, 256, **bf) for _ in range(2)]
        for n, N in enumerate(a.net):
            if n >= 2:
                break
            h = Y1[i][:, 256 + 512 * n:768 + 512 * n]
            N.in_, N.rows, N.ld_in, N.num_layers, N.elu_in = h.data_ptr(), M, 1280, 1, 1
            if n == 0:
                N.latent, N.wz, N.lat_ld, N.wz_ld, N.npv = lat.data_ptr(), wz.data_ptr(), 64, 64, 2
            L = N.layer[0]
            L.W, L.bias, L.out, L.n_out, L.k_in, L.ld_out, L.elu = W[n].data_ptr(), b[n].data_ptr(), outs[n].data_ptr() if store else None, 256, 512, 256, 0
        sets.append((a, outs))
    t = timeit([lambda q=q: lib.go1ppo_tail_fwd(ctypes.byref(q[0]), s) for q in sets])
    print(f"tail_fwd, one 512 -> 256 layer x 2 nets, rows/workgroup {'64' if os.environ.get('GO1PPO_TAIL_BM64') else '32'}, global stores {store}: {t:6.1f} us")
Z = [[torch.zeros(M, 256, **bf) for _ in range(2)] for _ in range(R)]
t_elu = timeit([lambda i=i: lib.go1ppo_elu_fwd(Y1[i][:, 256:].data_ptr(), M, 1024, 1280, lat.data_ptr(), 64, 2, wz.data_ptr(), 64, 512, s) for i in range(R)])
t_pair = timeit([lambda i=i: fused.gemm_nt_pair(lib, dict(a=Y1[i][:, 256:768], b=W[0], c=Z[i][0], bias=b[0]), dict(a=Y1[i][:, 768:], b=W[1], c=Z[i][1], bias=b[1]))
                 for i in range(R)])
print(f"current path: elu_fwd {t_elu:.1f} us + gemm_nt_pair {t_pair:.1f} us")
