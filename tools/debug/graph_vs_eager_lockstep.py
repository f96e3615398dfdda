"""Debug: eager and HIP-graph autograd updates in lockstep on identical rollouts; after every update, the first mini-batch step
whose result differs (weights after the step, running loss sums).   python tools/debug/graph_vs_eager_lockstep.py --envs 4096 --T 24"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(REPO, "walk-these-ways_amd")
for p in (os.path.join(PKG, "shims"), PKG, REPO):
    sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bf16", action="store_true")
    ap.add_argument("--updates", type=int, default=5)
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--T", type=int, default=24)
    args = ap.parse_args()
    from go1_gym_learn.ppo_cse.actor_critic import ActorCritic
    from go1_gym_learn.ppo_cse.ppo import PPO, PPO_Args
    N, T = args.envs, args.T
    PPO_Args.autocast_bf16, PPO_Args.use_fused_kernels = args.bf16, False
    algs, traces = [], []
    for use_graphs in (False, True):
        PPO_Args.use_hip_graphs = use_graphs
        torch.manual_seed(0)
        alg = PPO(ActorCritic(70, 2, 2100, 12), device="cuda:0")
        alg.init_storage(N, T, [70], [2], [2100], [12])
        trace = []
        for name in ("_minibatch_eager", "_minibatch_replay"):
            orig = getattr(alg, name)

            def wrapped(arg, _orig=orig, _alg=alg, _trace=trace):
                _orig(arg)
                if torch.cuda.is_current_stream_capturing():
                    return
                _trace.append((_alg.master.clone(), _alg._acc.clone(), _alg.master.grad.clone(), float(_alg._lr)))
            setattr(alg, name, wrapped)
        algs.append(alg)
        traces.append(trace)
    for it in range(args.updates):
        for k, alg in enumerate(algs):
            PPO_Args.use_hip_graphs = bool(k)
            g = torch.Generator(device="cuda").manual_seed(1000 + it)
            with torch.inference_mode():
                for t in range(T):
                    obs = torch.randn(N, 70, device="cuda", generator=g)
                    priv = torch.randn(N, 2, device="cuda", generator=g)
                    hist = torch.randn(N, 2100, device="cuda", generator=g)
                    torch.manual_seed(10 * it + t)
                    alg.act(obs, priv, hist)
                    alg.process_env_step(torch.randn(N, device="cuda", generator=g), torch.zeros(N, dtype=torch.uint8, device="cuda"),
                                         {"env_bins": torch.zeros(N, device="cuda"), "time_outs": torch.zeros(N, dtype=torch.bool, device="cuda")})
                alg.compute_returns(hist, priv)
            traces[k].clear()
            torch.manual_seed(100 + it)
            alg.update()
        st0, st1 = algs[0].storage, algs[1].storage
        same_in = all(torch.equal(getattr(st0, f), getattr(st1, f)) for f in ("observation_histories", "advantages", "returns", "values", "actions", "mu", "sigma", "actions_log_prob"))
        print(f"update {it}: inputs equal {same_in}  idx equal {torch.equal(algs[0]._idx_all, algs[1]._idx_all)}  graphs {len(algs[1]._graphs)}", flush=True)
        n = algs[0].n_body
        for s, ((w0, a0, g0, lr0), (w1, a1, g1, lr1)) in enumerate(zip(*traces)):
            if not torch.equal(w0, w1):
                pol = algs[0].policy
                worst = sorted(((float((pol._block(w0[:n], nm) - pol._block(w1[:n], nm)).abs().max()), nm) for nm, _ in pol.blocks), reverse=True)
                print(f"   first differing mini-batch step {s} (epoch {s // 4}, mini-batch {s % 4}): max|dw| {float((w0 - w1).abs().max()):.3e}  lr {lr0:.3e}/{lr1:.3e}", flush=True)
                print(f"   loss sums eager {a0.tolist()}  graph {a1.tolist()}", flush=True)
                print(f"   adaptation-stage gradient max diff {float((g0 - g1).abs().max()):.3e}  (|g| max {float(g0.abs().max()):.3e})", flush=True)
                print("   blocks: " + ", ".join(f"{nm} {d:.1e}" for d, nm in worst[:10] if d > 0) + f" | std {float((w0[n:n + 12] - w1[n:n + 12]).abs().max()):.1e}", flush=True)
                if s > 0:
                    (pw0, pa0, _, _), (pw1, pa1, _, _) = traces[0][s - 1], traces[1][s - 1]
                    print(f"   previous step: weights equal {torch.equal(pw0, pw1)}  loss sums equal {torch.equal(pa0, pa1)}", flush=True)
                return


if __name__ == "__main__":
    main()
