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
    args = ap.parse_args()
    from bench import build_env
    from go1_gym_learn.ppo_cse import Runner, RunnerArgs
    from go1_gym_learn.ppo_cse.ppo import PPO_Args
    PPO_Args.autocast_bf16 = not args.fp32
    RunnerArgs.save_video_interval = 0
    torch.manual_seed(0)
    env, cfg = build_env(args.envs, 0, 0)
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
                done = env.reset_buf.bool()
                acc[0] += env.rew_buf.sum(); acc[1] += n
                acc[2] += done.sum(); acc[3] += (done & env.time_out_buf).sum()
                ep_len_sum += (ep_before[done] + 1).sum()
            runner.alg.compute_returns(obs_dict["obs_history"][:n], obs_dict["privileged_obs"][:n])
        losses = runner.alg.update()
        if (it + 1) % args.every == 0:
            a = acc.tolist()
            print(f"it {it + 1:4d}  mean step reward {a[0] / a[1]:8.5f}  mean ep len {float(ep_len_sum) / max(a[2], 1):7.1f}  "
                  f"time-outs/resets {a[3] / max(a[2], 1):5.3f}  lr {runner.alg.learning_rate:.2e}  value loss {losses[0]:.4f}  "
                  f"surr {losses[1]:+.4f}  adapt {losses[2]:.4f}  std {float(runner.alg.std.mean()):.3f}  [{time.time() - t0:5.1f} s]", flush=True)
            acc.zero_(); ep_len_sum.zero_()


if __name__ == "__main__":
    main()
