"""Probe (round 6): does the rollout gain from two HALF batches stepping out of phase on two HIP streams?

The rollout is a strict chain per environment — step kernel (142 us, one master wavefront per CU, 84 % of the vector ALUs idle) -> inference +
storage (90 us of small kernels that do not fill the chip either) -> next step — so its two stages never overlap.  Two independent halves of 2048
environments, each a chain of its own on its own stream, could: the step kernel of half A (128 workgroups = 128 CUs) beside the inference of half B.
Built from existing pieces only: two complete environments + Runners of E / 2 environments (env_id_offset 0 and E / 2) against one of E, 24 rollout
steps + compute_returns each, no update.  GPU box only.

    python tools/probes/split_rollout_probe.py [--envs 4096]
"""
import argparse
import os
import sys
import time

R_ = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R_)
import bench  # noqa: E402
import torch  # noqa: E402


def make(envs, rank_offset, device):
    from go1_gym_learn.ppo_cse import Runner
    env, _ = bench.build_env(envs, 0, 0)
    runner = Runner(env, device=device)
    env.episode_length_buf.copy_(torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length)))
    return env, runner, env.get_observations()


def rollout(runner, env, od):
    T, n = runner.num_steps_per_env, env.num_train_envs
    with torch.inference_mode():
        for _ in range(T):
            od, _ = runner._rollout_step(od)
        runner.alg.compute_returns(od["obs_history"][:n], od["privileged_obs"][:n])
    runner.alg.storage.clear()
    return od


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--iters", type=int, default=30)
    args = ap.parse_args()
    device = "cuda:0"
    from go1_gym_learn.ppo_cse import RunnerArgs
    from go1_gym_learn.ppo_cse.ppo import PPO_Args
    PPO_Args.autocast_bf16 = True
    RunnerArgs.save_video_interval = 0
    E = args.envs
    print(f"# rollout only (24 env steps + compute_returns, storage cleared), {E} environments in all; ms per rollout, median of {args.iters}")
    # ---- one batch of E
    env, runner, od = make(E, 0, device)
    for _ in range(5):
        od = rollout(runner, env, od)
    torch.cuda.synchronize()
    ts = []
    for _ in range(args.iters):
        t0 = time.perf_counter()
        od = rollout(runner, env, od)
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    ts.sort()
    one = ts[len(ts) // 2]
    print(f"  one batch of {E}, one stream:                      {one:7.3f} ms  ({E * 24 / one / 1e3:6.2f} M env-steps/s)")
    del env, runner, od
    torch.cuda.empty_cache()
    # ---- two halves
    halves = [make(E // 2, i * (E // 2), device) for i in range(2)]
    streams = [torch.cuda.Stream(device=device) for _ in range(2)]

    def both(interleaved):
        T = halves[0][1].num_steps_per_env
        ods = [h[2] for h in halves]
        with torch.inference_mode():
            if interleaved:          # host issues A(t), B(t), A(t+1), ...: each half is a chain on its own stream
                for t in range(T):
                    for i, (env, runner, _) in enumerate(halves):
                        with torch.cuda.stream(streams[i]):
                            ods[i], _ = runner._rollout_step(ods[i])
            else:                    # the same two halves one after the other on ONE stream (what splitting alone costs)
                for t in range(T):
                    for i, (env, runner, _) in enumerate(halves):
                        ods[i], _ = runner._rollout_step(ods[i])
            for i, (env, runner, _) in enumerate(halves):
                n = env.num_train_envs
                with torch.cuda.stream(streams[i] if interleaved else torch.cuda.current_stream()):
                    runner.alg.compute_returns(ods[i]["obs_history"][:n], ods[i]["privileged_obs"][:n])
                runner.alg.storage.clear()
        for i in range(2):
            halves[i] = (halves[i][0], halves[i][1], ods[i])

    for interleaved in (False, True, False, True):
        for s in streams:
            s.wait_stream(torch.cuda.current_stream())
        for _ in range(5):
            both(interleaved)
        torch.cuda.synchronize()
        ts = []
        for _ in range(args.iters):
            t0 = time.perf_counter()
            both(interleaved)
            torch.cuda.synchronize()
            ts.append(1e3 * (time.perf_counter() - t0))
        ts.sort()
        med = ts[len(ts) // 2]
        what = "two halves out of phase on two streams:" if interleaved else "two halves in turn on one stream:      "
        print(f"  {what}            {med:7.3f} ms  ({E * 24 / med / 1e3:6.2f} M env-steps/s)   {one / med:5.3f} x the single batch's rate", flush=True)


if __name__ == "__main__":
    main()
