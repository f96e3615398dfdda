"""End-to-end learning check on the GPU box: train.py configuration, 4096 envs, N PPO iterations with the bench's
rollout/update loop; prints mean step reward, mean episode length of finished episodes, fraction of time-outs
among resets, and the learning rate every few iterations.  Usage: python tools/train_sanity.py [--iters 300]"""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "walk-these-ways_amd")
for p in (os.path.join(PKG, "shims"), PKG, REPO):
    sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=300)
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--every", type=int, default=20)
    ap.add_argument("--fp32", action="store_true")
    ap.add_argument("--no-graphs", action="store_true", help="PPO_Args.use_hip_graphs = False")
    ap.add_argument("--rough", action="store_true", help="BASELINE configs[2]: terrain-curriculum tile grid as a trimesh (vertical risers) + height scan")
    ap.add_argument("--spawn", default=None, choices=["tile_max", "centre_patch"], help="with --rough: spawn height of a tile (tile_max = the reference's rule)")
    ap.add_argument("--above-terrain", action="store_true", help="with --rough: rewards.heights_above_terrain (non-reference: foot / base heights of the reward terms above the ground)")
    ap.add_argument("--sigma-rew-neg", type=float, default=None, help="rewards.sigma_rew_neg (train.py: 0.02): reward = positive part x exp(negative part / sigma)")
    ap.add_argument("--save", default=None, help="write the policy (flat fp32 master parameter) to this file at the end")
    ap.add_argument("--check-finite", action="store_true", help="after every iteration: first non-finite tensor among "
                    "observations / rewards / actions / returns / parameters / gradients, then stop")
    args = ap.parse_args()
    from bench import build_env
    from go1_gym_learn.ppo_cse import Runner, RunnerArgs
    from go1_gym_learn.ppo_cse.ppo import PPO_Args
    PPO_Args.autocast_bf16 = not args.fp32
    if args.no_graphs:
        PPO_Args.use_hip_graphs = False
    RunnerArgs.save_video_interval = 0
    torch.manual_seed(0)
    env, cfg = build_env(args.envs, 0, 0, rough=args.rough, spawn=args.spawn, heights_above_terrain=args.above_terrain, sigma_rew_neg=args.sigma_rew_neg)
    per_class = None
    if args.rough:
        # terrain class of every environment's tile column (terrain.py make_terrain: choice = column / num_cols + 0.001 against the
        # cumulative proportions): which tile classes end episodes by falling
        base0 = env
        while not hasattr(base0, "terrain_types"):
            base0 = base0.env
        t = cfg.terrain
        cum = [sum(t.terrain_proportions[:i + 1]) for i in range(len(t.terrain_proportions))]
        names = ["slope (inverted: pit)", "slope", "rough slope", "stairs down (pit)", "stairs up", "obstacles"]

        def klass(col):
            c = col / t.num_cols + 0.001
            if c < cum[0]:
                return 0 if c < cum[0] / 2 else 1
            if c < cum[1]:
                return 2
            if c < cum[3]:
                return 3 if c < cum[2] else 4
            return 5
        cls = torch.tensor([klass(int(c)) for c in base0.terrain_types.tolist()], device="cuda")
        per_class = dict(names=names, cls=cls, stats=torch.zeros(len(names), 4, device="cuda"))      # episodes, time-outs, length sum, envs
        per_class["stats"][:, 3] = torch.bincount(cls, minlength=len(names)).float()
    runner = Runner(env, device="cuda:0")
    T = runner.num_steps_per_env
    buf = env.episode_length_buf
    buf.copy_(torch.randint_like(buf, high=int(env.max_episode_length)))
    obs_dict = env.get_observations()
    n = env.num_train_envs
    t0 = time.time()
    acc = torch.zeros(4, device="cuda")          # reward sum, steps, resets, time-outs
    ep_len_sum = torch.zeros((), device="cuda")
    for it in range(args.iters):
        with torch.inference_mode():
            for _ in range(T):
                ep_before = env.episode_length_buf.clone()
                obs_dict, infos = runner._rollout_step(obs_dict)
                if args.check_finite and not torch.isfinite(env.rew_buf).all():
                    base = env
                    while not hasattr(base, "buffers"):
                        base = base.env
                    B = base.buffers
                    e = int((~torch.isfinite(env.rew_buf)).nonzero()[0])
                    print(f"it {it + 1}: non-finite reward from the simulator, env {e}, reward {float(env.rew_buf[e])}", flush=True)
                    names = list(base.episode_sum_names)
                    for k, t in B.tensors.items():
                        if t is None or not t.is_floating_point() or t.shape[-1] != env.num_envs:
                            continue
                        col = t.reshape(-1, env.num_envs)[:, e].float().cpu()
                        if not torch.isfinite(col).all():
                            bad = (~torch.isfinite(col)).nonzero().flatten().tolist()
                            label = [names[i] if k in ("episode_sums", "command_sums") and i < len(names) else i for i in bad]
                            print(f"    {k}: non-finite rows {label}", flush=True)
                    for k in ("root_states", "commands", "contact_forces", "foot_positions", "foot_velocities", "dof_pos", "dof_vel", "torques",
                              "last_actions", "gait_indices", "desired_contact_states", "episode_length_buf"):
                        t = B.tensors.get(k)
                        if t is not None:
                            print(f"    {k} = {[round(float(v), 4) for v in t.reshape(-1, env.num_envs)[:, e].float().cpu()]}", flush=True)
                    return
                done = env.reset_buf.bool()
                acc[0] += env.rew_buf.sum(); acc[1] += n
                acc[2] += done.sum(); acc[3] += (done & env.time_out_buf).sum()
                ep_len_sum += (ep_before[done] + 1).sum()
                if per_class is not None and it >= args.iters - 100:          # the last 100 iterations
                    c, S = per_class["cls"][:n], per_class["stats"]
                    S[:, 0].index_add_(0, c, done[:n].float())
                    S[:, 1].index_add_(0, c, (done & env.time_out_buf)[:n].float())
                    S[:, 2].index_add_(0, c, torch.where(done[:n], (ep_before[:n] + 1).float(), torch.zeros(n, device="cuda")))
            runner.alg.compute_returns(obs_dict["obs_history"][:n], obs_dict["privileged_obs"][:n])
        if args.check_finite:
            st = runner.alg.storage
            pre = {k: getattr(st, k) for k in ("obs_ring" if st.ring else "observation_histories", "privileged_observations", "actions", "rewards", "values", "returns",
                                               "advantages", "mu", "sigma", "actions_log_prob")}
            bad = [k for k, t in pre.items() if not torch.isfinite(t.float()).all()]
            if bad:
                print(f"it {it + 1}: non-finite BEFORE the update in storage fields {bad}; master finite: "
                      f"{bool(torch.isfinite(runner.alg.master).all())}", flush=True)
                for k in bad:
                    t = pre[k].float()
                    nz = (~torch.isfinite(t)).nonzero()
                    print("   ", k, tuple(t.shape), "first bad index", nz[0].tolist(), "count", len(nz), flush=True)
                return
        losses = runner.alg.update()
        if args.check_finite:
            alg = runner.alg
            if not torch.isfinite(alg.master).all() or not all(map(lambda v: v == v, losses)):
                n = alg.n_body
                names = []
                for name, _ in alg.policy.blocks:
                    blk = alg.policy._block(alg.master[:n], name)
                    if not torch.isfinite(blk).all():
                        names.append(name)
                print(f"it {it + 1}: non-finite AFTER the update; losses {losses}; parameter blocks {names}; std finite "
                      f"{bool(torch.isfinite(alg.std).all())}", flush=True)
                net = alg._train_net
                if net is not None:
                    for label, t in (("Y1", net.Y1), ("dY1", net.dY1), ("Y1d", net.Y1d)):
                        print("   ", label, "finite:", bool(torch.isfinite(t.float()).all()), flush=True)
                    for nname, zs in net.Z.items():
                        for li, z in zs.items():
                            if not torch.isfinite(z.float()).all() or not torch.isfinite(net.dZ[nname][li].float()).all():
                                print("   ", nname, li, "Z finite", bool(torch.isfinite(z.float()).all()), "dZ finite",
                                      bool(torch.isfinite(net.dZ[nname][li].float()).all()), flush=True)
                return
        if (it + 1) % args.every == 0:
            base = env
            while not hasattr(base, "buffers"):
                base = base.env
            fl = base.extras["sim_faults"].consume()
            diag = (f"  | max|obs| {float(obs_dict['obs'].abs().max()):6.2f} max|v_base| {float(base.root_states[:, 7:10].norm(dim=1).max()):6.2f} "
                    f"max|qd| {float(base.dof_vel.abs().max()):5.1f} fallen {float((base.root_states[:, 2] < 0.15).float().mean()):.3f} "
                    f"fatal {fl['fatal']} dropped {fl.get('contact_dropped', 0)} {fl.get('contact_dropped_by_class', '')} limit_safety {fl.get('limit_safety', 0)}")
            a = acc.tolist()
            print(f"it {it + 1:4d}  mean step reward {a[0] / a[1]:8.5f}  mean ep len {float(ep_len_sum) / max(a[2], 1):7.1f}  "
                  f"time-outs/resets {a[3] / max(a[2], 1):5.3f}  lr {runner.alg.learning_rate:.2e}  value loss {losses[0]:.4f}  "
                  f"surr {losses[1]:+.4f}  adapt {losses[2]:.4f}  std {float(runner.alg.std.detach().mean()):.3f}  [{time.time() - t0:5.1f} s]{diag}", flush=True)
            acc.zero_(); ep_len_sum.zero_()
    # per-term episode sums of the episodes finished since the last print (reference extras["train/episode"], legged_robot.py:181-227):
    # which terms make up train.py's reward = positive part x exp(negative part / sigma_rew_neg)
    base = env
    while not hasattr(base, "buffers"):
        base = base.env
    terms = {k: v for k, v in base.extras["train/episode"].consume().items() if k.startswith("rew_")}
    print("mean per-episode reward terms (finished episodes since the last print), most negative first:")
    for k, v in sorted(terms.items(), key=lambda kv: kv[1])[:8]:
        print(f"    {k:44s} {v:12.4f}")
    for k, v in sorted(terms.items(), key=lambda kv: -kv[1])[:4]:
        print(f"    {k:44s} {v:12.4f}")
    if args.save:
        torch.save(runner.alg.sync_module().state_dict(), args.save)
    if per_class is not None:
        print("per terrain class, last 100 iterations: envs | episodes ended | time-outs among them | mean episode length [steps]")
        for name, (ep, to, ln, ne) in zip(per_class["names"], per_class["stats"].tolist()):
            print(f"  {name:24s} {int(ne):5d} | {int(ep):7d} | {to / max(ep, 1):5.3f} | {ln / max(ep, 1):7.1f}")


if __name__ == "__main__":
    main()
