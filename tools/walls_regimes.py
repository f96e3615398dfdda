#!/usr/bin/env python3
"""go1_step_kernel_walls: launch time and contact load in the regimes the round-5 numbers were quoted on (the review of round 5, item 1b).

    python tools/walls_regimes.py [--envs 4096] > profiles/r06_walls_regimes.txt

Regimes, each on a FRESH BASELINE configs[2] environment (bench.build_env(rough=True)), N(0,1) actions in the timed window as in
bench.time_sim_only (48 warm-up + 240 timed launches, HIP events of the library):
  fresh        `bench.py --sim-only --rough`'s regime: the timing starts on the reset distribution
  aged         the same after 168 more N(0,1) steps (the env-steps the bench leg's 7 iterations take), no policy
  bench-leg    what bench.py's `rough_trimesh` leg does: 7 live-policy PPO iterations (2 warm-up + 5 timed), THEN the sim-only window
  hot          fresh environment, but the device has just run 20 s of back-to-back bf16 GEMMs (the state bench.py's leg meets: it comes after the
               headline's 223 PPO iterations) — same workload, same contact load: what is left is the device's clock / power state
  +churn       (with after-flat) three more flat environments + Runners built, run for 3 iterations and dropped first: the allocator's free lists
               as bench.py's rough_trimesh leg meets them (after `dropin_default`)
  after-flat   the rough environment built in a process that already holds the headline's flat environment + Runner and ran 5 of its PPO
               iterations (what bench.py's leg meets); `+empty_cache`: the same with torch.cuda.empty_cache() before the rough environment
               is built (allocator state: are the environment's ~100 SoA arrays carved out of recycled segments?)
Contact load: the `_sig` twin of the kernel stepped over a copy of the state right after the timed window, 24 steps; per environment and
substep the listed top-surface points (signature word 0), wall + hip points (word 1; bits 0..12 = wall points), self contacts (word 2).
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
import bench  # noqa: E402  (puts the package on sys.path)
import torch  # noqa: E402


def popcount(t):
    t = t.to(torch.int64) & 0xFFFFFFFF
    c = torch.zeros_like(t)
    for b in range(32):
        c += (t >> b) & 1
    return c


def contact_load(env, steps=24):
    import go1sim_host as H
    import go1sim_abi as abi
    inner = env.env
    B = inner.buffers
    Bt = B.clone_to(B.device)
    Bt.enable_contact_signature()
    sim = H.Go1Sim(inner.sim_config, Bt, inner.sim_device_id)
    if inner.sim_config_eval is not None:
        sim.set_eval_config(inner.sim_config_eval, inner.num_train_envs)
    sim.set_counters(*inner.sim.counters())
    n = B.root_states.shape[1]
    top = wall = hip = selfc = 0.0
    mx = 0
    resets = 0
    W = abi.GO1_SIG_WORDS
    for t in range(steps):
        sim.step(torch.randn(n, 12, device=B.device))
        sig = Bt.contact_signature.view(abi.GO1_SIG_MAX_SUBSTEPS, W, n)
        a = popcount(sig[:, 0])
        w = popcount(sig[:, 1] & 0x1FFF)
        h = popcount(sig[:, 1] & ~0x1FFF)
        w2 = sig[:, 2].to(torch.int64) & 0xFFFFFFFF
        s = sum(((w2 >> (3 * p)) & 7 != 0).long() for p in range(6)) + popcount((w2 >> 18) & 0xF)
        top += float(a.float().mean())
        wall += float(w.float().mean())
        hip += float(h.float().mean())
        selfc += float(s.float().mean())
        mx = max(mx, int((a + w + h + s).max()))
        resets += int(Bt.reset_buf.sum())
    k = float(steps)
    return dict(top=top / k, wall=wall / k, hip=hip / k, self=selfc / k, listed=(top + wall + hip + selfc) / k, max_listed=mx,
                resets_per_step=resets / k)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--regimes", default="fresh,after-flat,after-flat+empty_cache,after-flat+bench-leg,fresh",
                    help="comma-separated, in order: fresh, aged, bench-leg, hot, after-flat[+empty_cache][+bench-leg]")
    args = ap.parse_args()
    device = "cuda:0"
    from go1_gym_learn.ppo_cse import Runner, RunnerArgs
    from go1_gym_learn.ppo_cse.ppo import PPO_Args
    PPO_Args.autocast_bf16 = True
    RunnerArgs.save_video_interval = 0
    print(f"# go1_step_kernel_walls at {args.envs} envs, BASELINE configs[2]; launch ms = HIP events around the launch (go1sim_read_timings), 240 launches")
    print("# regime      launch_ms  sim_only_M/s  listed/env/substep (top + wall + hip + self)   max   resets/step")
    keep = None
    for regime in args.regimes.split(","):
        torch.manual_seed(0)
        if regime.startswith("after-flat") and keep is None:
            envf, _ = bench.build_env(args.envs, 0, 0)
            runf = Runner(envf, device=device)
            envf.episode_length_buf.copy_(torch.randint_like(envf.episode_length_buf, high=int(envf.max_episode_length)))
            _, odf = bench.time_iterations(runf, envf, envf.get_observations(), 5, warmup=2)
            keep = (envf, runf, odf)
        if "churn" in regime:             # what bench.py's later legs meet: other environments + Runners were built, run and dropped before
            for _ in range(3):
                e2, _ = bench.build_env(args.envs, 0, 0)
                r2 = Runner(e2, device=device)
                e2.episode_length_buf.copy_(torch.randint_like(e2.episode_length_buf, high=int(e2.max_episode_length)))
                bench.time_iterations(r2, e2, e2.get_observations(), 2, warmup=1)
                del e2, r2
        if "empty_cache" in regime:
            torch.cuda.empty_cache()
        env, _ = bench.build_env(args.envs, 0, 0, rough=True)
        env.episode_length_buf.copy_(torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length)))       # as bench.py main() and its leg do
        if regime.endswith("bench-leg") or regime.endswith("bench-leg-dephased"):
            runner = Runner(env, device=device)          # (Runner.__init__ resets the environment: the episode lengths are back at 0 ...)
            if regime.endswith("dephased"):              # (... unless they are spread again AFTER it, which is the order bench.py's leg has)
                env.episode_length_buf.copy_(torch.randint_like(env.episode_length_buf, high=int(env.max_episode_length)))
            od = env.get_observations()
            _, od = bench.time_iterations(runner, env, od, 5, warmup=2)
        elif regime == "hot":
            a = torch.randn(8192, 8192, device=device, dtype=torch.bfloat16)
            b = torch.randn(8192, 8192, device=device, dtype=torch.bfloat16)
            c = torch.empty_like(a)
            import time
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 20.0:
                for _ in range(50):
                    torch.mm(a, b, out=c)
                torch.cuda.synchronize()
            del a, b, c
        elif regime == "aged":
            for _ in range(168):
                env.step(torch.randn(args.envs, 12, device=device))
        rate, ms = bench.time_sim_only(env, env.env.sim, args.envs, 240, device)
        c = contact_load(env)
        sys.stdout.flush()
        print(f"  {regime:24s}  {ms:8.4f}  {rate / 1e6:10.2f}     {c['listed']:.2f} ({c['top']:.2f} + {c['wall']:.3f} + {c['hip']:.3f} + {c['self']:.3f})"
              f"            {c['max_listed']:3d}   {c['resets_per_step']:.1f}")
        del env
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
