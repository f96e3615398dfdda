"""Per-phase cycle breakdown of go1_step_kernel (one wave: workgroup 0, lane 0), using the -DGO1_PROFILE build.
GPU box:  python tools/phase_profile.py [--envs 4096] [--steps 50] [extra hipcc flags...]
Rebuilds csrc/libgo1sim.so with the markers, runs `steps` env steps of the train.py configuration with N(0,1)
actions, prints accumulated s_memtime deltas per phase, then restores the normal library."""
import argparse
import ctypes
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "walk-these-ways_amd")
CSRC = os.path.join(PKG, "csrc")
for p in (os.path.join(PKG, "shims"), PKG, REPO):
    sys.path.insert(0, p)

PHASES = ["rest of the prologue (stash commit, input checks)", "torque model: barrier after the publish (x4)", "torque pick-up (barrier wait) + ABA pass 2 (x4)",
          "prologue: the load batch lands + warm-start impulses into LDS", "ABA pass 3 + limit-row bounds + hand-over packets + barrier (x4)", "PGS sweeps (x4)", "apply + integrate (x4)",
          "store state/feet/forces", "post: derived + callbacks", "post: gait clock + push/dof-rand",
          "post: feet/heights/termination", "post: rewards", "post: reset", "post: observations", "post: privileged obs",
          "post: roll", "post: reward-input loads", "post: termination", "fault flags + sweep bounds (x4)",
          "kinematics + candidates + self-collision geometry + contact list (x4)", "post items into the records (x4)",
          "self-contacts + limit rows + barrier wait for the helpers' rows (x4)",
          "torque model: post q, qd for the helpers' rows (x4)", "PGS: warm-start state (x4)",
          "loop edge after the pose update (x4)", "kinematics + terrain candidates (x4; 19 = the contact list alone)", "self-collision geometry (x4)",
          "apply: impulses of the listed contacts from LDS (x4; 6 = the rest)", "apply: back-substitution + base twist (x4)", "integrate the joints (x4)",
          "what the compiler sank behind the substep's last marker (x4; 24 = the loop's back edge alone)",
          "sweep: cooperative turns (x16; 5 = the sweep's rest: signature, phase marker)", "sweep: the legs' turns + merge (x16)", "sweep: limit rows (x16)"]


def build(flags):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize"] + flags + \
          ["-o", os.path.join(CSRC, "libgo1sim.so"), os.path.join(CSRC, "go1sim.hip")]
    subprocess.check_call(cmd, cwd=CSRC)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--lib", default=None, help="a library prebuilt with -DGO1_PROFILE (no rebuild on the GPU box, csrc/libgo1sim.so untouched)")
    ap.add_argument("--zero-actions", action="store_true", help="robots standing on four feet instead of N(0,1) actions")
    ap.add_argument("--rough", action="store_true", help="BASELINE configs[2]: the terrain-curriculum tile grid as a trimesh terrain + height scan "
                                                         "(go1_step_kernel_walls; the markers of workgroup 0 and the per-workgroup statistics)")
    ap.add_argument("--aged", default=None, choices=["inphase", "dephased"],
                    help="with --rough: first the 7 live-policy PPO iterations of bench.py's rough_trimesh leg — `dephased`: episode lengths spread over the "
                         "horizon AFTER the Runner's reset, as bench.py has them; `inphase`: all episodes started together (profiles/r06_walls_regimes.txt)")
    args, extra = ap.parse_known_args()
    if args.lib is None:
        build(["-DGO1_PROFILE"] + extra)
    try:
        import torch
        import go1sim_host as H
        if args.lib is not None:
            H.LIB_PATH, H._lib = os.path.abspath(args.lib), None
        from go1_gym.envs.base.legged_robot_config import make_cfg
        from go1_gym.envs.go1.velocity_tracking import VelocityTrackingEasyEnv
        from scripts.train_config import apply_train_config
        if args.rough:
            sys.path.insert(0, REPO)
            from bench import build_env
            wrapped, cfg = build_env(args.envs, 0, 0, rough=True)
            env = wrapped.env if hasattr(wrapped, "env") else wrapped
            env.reset()
            if args.aged:
                import bench
                from go1_gym_learn.ppo_cse import Runner, RunnerArgs
                from go1_gym_learn.ppo_cse.ppo import PPO_Args
                PPO_Args.autocast_bf16, RunnerArgs.save_video_interval = True, 0
                runner = Runner(wrapped, device="cuda:0")
                if args.aged == "dephased":
                    wrapped.episode_length_buf.copy_(torch.randint_like(wrapped.episode_length_buf, high=int(wrapped.max_episode_length)))
                bench.time_iterations(runner, wrapped, wrapped.get_observations(), 5, warmup=2)
        else:
            cfg = apply_train_config(make_cfg(), num_envs=args.envs)
            env = VelocityTrackingEasyEnv(sim_device="cuda:0", headless=True, cfg=cfg)
        lib = H.load_library()
        lib.go1sim_debug_read_profile.argtypes = [ctypes.c_void_p]
        buf = (ctypes.c_uint64 * 64)()
        acts = (torch.zeros if args.zero_actions else torch.randn)(args.steps, args.envs, 12, device="cuda")
        for t in range(10):
            env.step(acts[t])
        torch.cuda.synchronize()
        lib.go1sim_debug_read_profile(buf)            # clear
        wg = (ctypes.c_uint64 * 1152)()
        lib.go1sim_debug_read_wg_times.argtypes = [ctypes.c_void_p]
        lib.go1sim_debug_read_wg_times(wg)
        wp = (ctypes.c_uint64 * (1024 * 40))()
        lib.go1sim_debug_read_wg_phases.argtypes = [ctypes.c_void_p]
        lib.go1sim_debug_read_wg_phases(wp)
        for t in range(args.steps):
            env.step(acts[t])
        torch.cuda.synchronize()
        assert lib.go1sim_debug_read_profile(buf) == 0
        assert lib.go1sim_debug_read_wg_times(wg) == 0
        import numpy as np
        w = np.array(wg[:args.envs // 16], dtype=np.float64) / args.steps
        print(f"master wavefront's cycles per step over the {len(w)} workgroups: mean {w.mean():.0f}, median {np.median(w):.0f}, 90 % {np.quantile(w, 0.9):.0f}, "
              f"99 % {np.quantile(w, 0.99):.0f}, max {w.max():.0f} (the launch lasts as long as its slowest workgroup: max / mean = {w.max() / w.mean():.2f})")
        nwg = args.envs // 16
        lmax, lsum = np.array(wg[1024:1088], dtype=np.float64), np.array(wg[1088:1152], dtype=np.float64)
        used = lmax > 0
        print(f"per launch ({int(used.sum())} launches): mean over workgroups {np.mean(lsum[used] / nwg):.0f} cycles, slowest workgroup {np.mean(lmax[used]):.0f} "
              f"(ratio {np.mean(lmax[used] / (lsum[used] / nwg)):.2f}: what perfectly even workgroups would save)")
        tot = sum(buf[:len(PHASES)])
        print(f"cycles per step (wave 0 lane 0, s_memtime ticks): {tot / args.steps:.0f}")
        assert lib.go1sim_debug_read_wg_phases(wp) == 0
        P = np.array(wp[:], dtype=np.float64).reshape(1024, 40)[:nwg, :len(PHASES)] / args.steps
        order = np.argsort(P.sum(1))
        slow, fast = P[order[-max(nwg // 10, 1):]].mean(0), P[order[:max(nwg // 10, 1)]].mean(0)
        print(f"  phase                                      workgroup 0    mean of all   slowest 10 %   fastest 10 %   (cycles per step)")
        for i, name in enumerate(PHASES):
            print(f"  {i:2d} {name[:40]:40s} {buf[i] / args.steps:10.0f} {P[:, i].mean():12.0f} {slow[i]:14.0f} {fast[i]:14.0f}")
    finally:
        if args.lib is None:
            build([])


if __name__ == "__main__":
    main()
