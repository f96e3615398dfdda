"""Probe (round 6): where do the PPO loss kernel's 22 us go?  libgo1ppo.so built with -DLOSS_ABLATE=<bits> (csrc/go1ppo.hip: 1 = storage rows in
order instead of through the mini-batch permutation, 2 = hardware log / exp instead of the correctly rounded ones, 4 = no d_mean / d_value
stores, 8 = the block reductions without their atomics), each timed on the update's shapes — 24576 rows of a 98304-sample storage — in both
forms of the sums (fp32 atomics from every workgroup; workspace rows + ticket).  Build: see the loop in the docstring of this file's commit,
    for v in 0 1 2 4 8 15; do hipcc ... -DLOSS_ABLATE=$v -o walk-these-ways_amd/csrc/variants/ppo_loss_ablate_$v.so go1ppo.hip; done
GPU box only."""
import glob
import os
import sys

R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P_ = os.path.join(R_, "walk-these-ways_amd")
for p in (os.path.join(P_, "shims"), P_, R_):
    sys.path.insert(0, p)
import torch  # noqa: E402
from go1_gym_learn.ppo_cse import fused  # noqa: E402


def timeit(fn, iters=200, warm=20):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    R, M, na = 98304, 24576, 12
    g = torch.Generator(device="cuda").manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device="cuda", generator=g)
    store = dict(actions=rnd(R, na), mu=rnd(R, na) * 0.3, sigma=torch.rand(R, na, device="cuda", generator=g) + 0.5, logp=rnd(R) * 0.5 - 14.0,
                 adv=rnd(R), returns=rnd(R), values=rnd(R))
    idx = torch.randperm(R, device="cuda", generator=g)[:M].contiguous()
    mean = torch.zeros(M, 64, device="cuda", dtype=torch.bfloat16)
    mean[:, :na] = (store["mu"][idx] + 0.05 * rnd(M, na)).to(torch.bfloat16)
    value = torch.zeros(M, 64, device="cuda", dtype=torch.bfloat16)
    value[:, 0] = store["values"][idx].to(torch.bfloat16)
    std = torch.rand(na, device="cuda", generator=g) + 0.6
    dmean, dvalue = torch.zeros_like(mean), torch.zeros_like(value)
    out = torch.zeros(64, device="cuda")
    # something else between two launches, as in the update (the loss kernel never runs back to back): flush the caches with a 256 MB copy
    big_a = torch.empty(64 << 20, device="cuda", dtype=torch.float32)
    big_b = torch.empty_like(big_a)
    print(f"# go1ppo_loss, {M} rows of a {R}-sample storage, {na} actions; us per launch (cold: a 256 MB copy between launches, its time subtracted)")
    libs = sorted(glob.glob(os.path.join(P_, "csrc", "variants", "ppo_loss_ablate_*.so")), key=lambda p: int(p.split("_")[-1][:-3]))
    s = torch.cuda.current_stream().cuda_stream
    t_copy = timeit(lambda: big_b.copy_(big_a), iters=30, warm=3)
    for path in libs:
        lib = fused.load_library(path)
        bits = int(path.split("_")[-1][:-3])
        res = []
        for ws_on in ((False, True) if hasattr(fused.LossArgs, "workspace") else (False,)):     # (the workspace form was withdrawn: r06_loss_kernel_ablation.txt)
            a = fused.LossArgs()
            a.mean, a.value, a.std, a.head_ld, a.num_actions, a.rows = mean.data_ptr(), value.data_ptr(), std.data_ptr(), 64, na, M
            a.idx = idx.data_ptr()
            a.actions, a.old_mu, a.old_sigma = store["actions"].data_ptr(), store["mu"].data_ptr(), store["sigma"].data_ptr()
            a.old_logp, a.advantages, a.returns, a.old_values = (store[k].data_ptr() for k in ("logp", "adv", "returns", "values"))
            a.clip_param, a.value_loss_coef, a.entropy_coef, a.use_clipped_value_loss = 0.2, 1.0, 0.01, 1
            a.d_mean, a.d_value = dmean.data_ptr(), dvalue.data_ptr()
            a.surrogate_loss, a.value_loss, a.kl = out[0:].data_ptr(), out[1:].data_ptr(), out[2:].data_ptr()
            a.d_std, a.d_mean_bias, a.d_value_bias = out[3:].data_ptr(), out[3 + na:].data_ptr(), out[3 + 2 * na:].data_ptr()
            if ws_on:
                ws = torch.zeros(int(lib.go1ppo_loss_workspace_bytes(M)) // 4, device="cuda")
                a.workspace = ws.data_ptr()
            hot = timeit(lambda: lib.go1ppo_loss(a, s))
            cold = timeit(lambda: (big_b.copy_(big_a), lib.go1ppo_loss(a, s)), iters=30, warm=3) - t_copy
            res.append((hot, cold))
        what = " + ".join(n for b, n in ((1, "rows in order"), (2, "hardware log/exp"), (4, "no stores"), (8, "no atomics")) if bits & b) or "product"
        ws_txt = f" | workspace {res[1][0]:6.1f} hot {res[1][1]:6.1f} cold" if len(res) > 1 else ""
        print(f"  ablate {bits:2d} ({what:55s}): atomics {res[0][0]:6.1f} hot {res[0][1]:6.1f} cold{ws_txt}")


if __name__ == "__main__":
    main()
