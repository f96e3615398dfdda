"""Task-level acceptance of the simulator (the reference's physics cannot be pinned — it is the closed Isaac Gym binary —, so the
substitute is behavioural): train the reference's algorithm on this simulator with scripts/train.py's configuration, then evaluate
the policy the way scripts/play.py does (reference play.py:89-139: commands written every step — x velocity, trotting gait at
3 Hz, 0.08 m footswing, 0.25 m stance width —, deterministic actions, 250 steps), over many environments instead of one:

  * velocity tracking      mean |v_x - v_cmd| over the last 150 steps (base frame, env.base_lin_vel as play.py plots it)
  * heading                |yaw drift| over the 250 steps (commanded yaw rate 0)
  * gait                   fraction of (step, foot) samples where the measured contact (F_z > 1 N) equals the commanded
                           contact schedule (desired_contact_states > 0.5) — the trot the gait command asks for
  * falls                  fraction of environments whose episode terminated during the evaluation

    python tools/play_eval.py --iters 5000 [--eval-at 1500 3000 5000] [--vx 1.0 1.5]

Run on the GPU box; the summary goes to stdout (committed as profiles/r03_play_eval.txt)."""
import argparse
import math
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "walk-these-ways_amd")
for p in (os.path.join(PKG, "shims"), PKG, REPO):
    sys.path.insert(0, p)
import torch  # noqa: E402


def yaw_of(quat_xyzw):
    x, y, z, w = quat_xyzw.unbind(-1)
    return torch.atan2(2 * (w * z + x * y), 1 - 2 * (y * y + z * z))


def evaluate(runner, num_envs, vx, steps=250, seed=1, solver_sweeps=None):
    """a FRESH evaluation environment (play.py builds its own): DR ranges of the training configuration, commands written
    every step, deterministic policy (actor mean on the adaptation module's latent: act_student's inference path)"""
    from bench import build_env
    env, cfg = build_env(num_envs, 0, seed, solver_sweeps=solver_sweeps)
    base = env.env
    alg = runner.alg
    policy = alg.actor_critic
    alg.sync_module()
    policy.eval()
    obs = env.reset()
    obs = env.get_observations()
    cmd = torch.zeros(num_envs, base.commands.shape[1], device=base.device)
    cmd[:, 0], cmd[:, 4], cmd[:, 5], cmd[:, 8], cmd[:, 9], cmd[:, 12] = vx, 3.0, 0.5, 0.5, 0.08, 0.25       # trot: phase 0.5, offsets 0
    if cmd.shape[1] > 13:
        cmd[:, 13] = 0.40                                       # stance length (train.py range [0.35, 0.45])
    yaw0 = None
    fell = torch.zeros(num_envs, dtype=torch.bool, device=base.device)
    verr, match, n_match = 0.0, 0.0, 0
    with torch.inference_mode():
        for i in range(steps):
            actions = policy.act_inference(obs) if hasattr(policy, "act_inference") else policy.act_student(obs["obs_history"])
            base.commands[:] = cmd
            obs, rew, done, info = env.step(actions)
            if i == 0:
                yaw0 = yaw_of(base.root_states[:, 3:7]).clone()
            fell |= done.bool() & ~base.time_out_buf.bool()
            if i >= steps - 150:
                verr += float((base.base_lin_vel[:, 0] - vx).abs()[~fell].mean()) / 150
                contact = base.contact_forces[:, base.feet_indices, 2] > 1.0
                want = base.desired_contact_states > 0.5
                match += float((contact == want)[~fell].float().mean())
                n_match += 1
        dyaw = yaw_of(base.root_states[:, 3:7]) - yaw0
        dyaw = torch.atan2(torch.sin(dyaw), torch.cos(dyaw)).abs()
    return dict(vx=vx, vel_err=verr, yaw_drift=float(dyaw[~fell].mean()) if bool((~fell).any()) else float("nan"),
                gait_match=match / max(n_match, 1), fall_rate=float(fell.float().mean()),
                mean_vx=float(base.base_lin_vel[:, 0][~fell].mean()) if bool((~fell).any()) else float("nan"))


def train_and_evaluate(iters, envs=4096, eval_envs=512, eval_at=None, vxs=(1.0, 1.5), log_every=500, fp32=False, out=print, after=None,
                       sweeps=None, eval_sweeps=(None,)):
    """Train with scripts/train.py's configuration (bench.py's loop), evaluate at the iterations `eval_at`; returns
    ({iteration: [evaluate() records]}, fault totals over the training).  `after(runner, env, obs_dict)`: called once the training is over."""
    from bench import build_env
    from go1_gym_learn.ppo_cse import Runner, RunnerArgs
    from go1_gym_learn.ppo_cse.ppo import PPO_Args
    saved = PPO_Args.autocast_bf16
    PPO_Args.autocast_bf16 = not fp32
    RunnerArgs.save_video_interval = 0
    torch.manual_seed(0)
    env, cfg = build_env(envs, 0, 0, solver_sweeps=sweeps)
    runner = Runner(env, device="cuda:0")
    PPO_Args.autocast_bf16 = saved
    T = runner.num_steps_per_env
    env.episode_length_buf.copy_(torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length)))
    obs_dict = env.get_observations()
    n = env.num_train_envs
    eval_at = sorted(set(eval_at or [iters]))
    t0 = time.time()
    rew_acc, cnt = torch.zeros((), device="cuda"), 0
    totals, results = {}, {}
    out(f"# train.py configuration, {envs} envs, ppo_cse, {'fp32' if fp32 else 'bf16'} policy; evaluation: {eval_envs} fresh environments, play.py commands")
    for it in range(1, iters + 1):
        with torch.inference_mode():
            for _ in range(T):
                obs_dict, _ = runner._rollout_step(obs_dict)
                rew_acc += env.rew_buf.mean(); cnt += 1
            runner.alg.compute_returns(obs_dict["obs_history"][:n], obs_dict["privileged_obs"][:n])
        runner.alg.update()
        if it % log_every == 0 or it == iters:
            faults = env.env.extras["sim_faults"].consume()
            for k, v in faults.items():
                if isinstance(v, dict):
                    for kk, vv in v.items():
                        totals[f"{k}.{kk}"] = totals.get(f"{k}.{kk}", 0) + int(vv)
                else:
                    totals[k] = totals.get(k, 0) + int(v)
            out(f"it {it:5d}  {it * T * envs / 1e6:7.1f} M env-steps  mean step reward {float(rew_acc) / max(cnt, 1):8.5f}  lr {runner.alg.learning_rate:.2e}  "
                f"std {float(runner.alg.std.detach().mean()):.3f}  sim faults fatal {faults['fatal']} dropped {faults.get('contact_dropped', 0)}  [{time.time() - t0:6.1f} s]")
            rew_acc.zero_(); cnt = 0
        if it in eval_at:
            results[it] = []
            for vx, es in [(v, e) for e in eval_sweeps for v in vxs]:
                r = evaluate(runner, eval_envs, vx, solver_sweeps=es)
                r["solver_sweeps"] = es
                results[it].append(r)
                out(f"EVAL it {it:5d}  {'' if es is None else f'[{es} solver sweeps] '}v_cmd {vx:.1f}: mean v_x {r['mean_vx']:.3f}  |v_x - v_cmd| {r['vel_err']:.3f} m/s  yaw drift {r['yaw_drift']:.3f} rad  "
                    f"gait-schedule match {r['gait_match']:.3f}  fall rate {r['fall_rate']:.3f}")
            runner.alg.actor_critic.train()
    if after is not None:
        after(runner, env, obs_dict)
    steps = iters * T * envs
    out(f"FAULT SOAK over {steps / 1e6:.1f} M env-steps of training: " + "  ".join(f"{k} {v} ({v / steps:.2e}/env-step)" for k, v in sorted(totals.items())))
    return results, totals


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5000)
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--eval-envs", type=int, default=512)
    ap.add_argument("--eval-at", type=int, nargs="*", default=None)
    ap.add_argument("--vx", type=float, nargs="*", default=[1.0, 1.5])
    ap.add_argument("--log-every", type=int, default=500)
    ap.add_argument("--fp32", action="store_true", help="the autograd fp32 update instead of the bf16 fused one (bench.py's configuration)")
    ap.add_argument("--sweeps", type=int, default=None, help="PGS sweeps per substep of the TRAINING simulator (default: the reference's 4)")
    ap.add_argument("--eval-sweeps", type=int, nargs="*", default=None,
                    help="evaluate the same policy on simulators with these sweep counts (solver-convergence sensitivity of the task metrics)")
    args = ap.parse_args()
    if args.sweeps is not None:
        print(f"# training simulator: {args.sweeps} solver sweeps per substep", flush=True)
    train_and_evaluate(args.iters, args.envs, args.eval_envs, args.eval_at, args.vx, args.log_every, args.fp32,
                       out=lambda m: print(m, flush=True), sweeps=args.sweeps, eval_sweeps=tuple(args.eval_sweeps or (None,)))


if __name__ == "__main__":
    main()
