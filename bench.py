#!/usr/bin/env python3
"""bench.py — env-steps/sec (sim+PPO) at 4096 Go1 envs per GPU (BASELINE.json metric), MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One *step* = one PPO iteration of the hot path over a fresh batch: 24 policy steps x 4096 envs through the fused
HIP step kernel (torque model + 4 physics substeps + tensor maps each), policy inference for every step
(`alg.act`), storage, GAE, then the full update (5 epochs x 4 mini-batches x {PPO step, adaptation step}).
Workload = BASELINE.json configs[1]: Go1 flat terrain, 4096 envs, train.py configuration (actuator net, lag 6,
domain randomisation, device command curriculum), bf16 policy.  Synthetic: random-init policy acting in closed
loop, initial states from the reset distribution.  N > 1: every rank owns 4096 envs (weak scaling), gradients are
all-reduced over RCCL; value = total env-steps of all ranks / max-over-ranks time.

Extra JSON objects: `roofline` (step kernel's algorithmic HBM bytes / per-launch HIP-event duration, DESIGN.md
"Measurement") and, on rank 0 at N=1, `cpu_baseline` (the fp64 oracle restatement of the same env step on the
host cores — the reference's own CPU path, Isaac Gym CPU-PhysX, is a closed binary that cannot run here).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(REPO, "walk-these-ways_amd")
for p in (os.path.join(PKG, "shims"), PKG, REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

# GEMM kernel selection for the PPO update: PPO() enables PyTorch TunableOp with the gfx950 table shipped in
# walk-these-ways_amd/tuning/ (go1_gym_learn/ppo_cse/ppo.py::_enable_tuned_gemms; every rank reads the same file;
# shapes missing from the table are tuned once during the warm-up iterations).  PYTORCH_TUNABLEOP_* overrides it.

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

ALGORITHMIC_BYTES_PER_ENV_STEP = 3444      # 1,332 B read + 1,832 B written + 280 B history append
HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def build_env(num_envs, rank, seed, rough=False, curriculum_update_interval=None, spawn=None, heights_above_terrain=False, sigma_rew_neg=None, solver_sweeps=None):
    """train.py configuration (BASELINE configs[1]); rough=True: configs[2] — the terrain curriculum's tile grid
    (slopes / rough slopes / stairs / obstacles, cfg:64-102 defaults) as a trimesh terrain (vertical risers) + the 187-point
    height scan appended to the observation (70 + 187 = 257)."""
    from go1_gym.envs.base.legged_robot_config import make_cfg
    from go1_gym.envs.go1.velocity_tracking import VelocityTrackingEasyEnv
    from go1_gym.envs.wrappers.history_wrapper import HistoryWrapper
    from scripts.train_config import apply_train_config
    cfg = apply_train_config(make_cfg(), num_envs=num_envs)
    cfg.seed = seed
    cfg.env.env_id_offset = rank * num_envs
    if curriculum_update_interval is not None:
        cfg.commands.curriculum_update_interval = int(curriculum_update_interval)
    if solver_sweeps is not None:                # diagnostic (tools/play_eval.py --sweeps): the reference's PhysX setting is 4
        cfg.sim.physx.num_solver_sweeps = int(solver_sweeps)
    if sigma_rew_neg is not None:                # diagnostic (tools/train_sanity.py): train.py's value is 0.02
        cfg.rewards.sigma_rew_neg = float(sigma_rew_neg)
    if rough:
        t = cfg.terrain
        # 'trimesh' with the default slope_treshold 0.75: stair risers / obstacle sides are VERTICAL faces (terrain.py:33-36)
        t.mesh_type, t.terrain_proportions, t.curriculum = "trimesh", [0.1, 0.1, 0.35, 0.25, 0.2], True
        t.num_rows, t.num_cols, t.terrain_length, t.terrain_width, t.border_size, t.center_robots = 10, 20, 8.0, 8.0, 25.0, False
        t.measure_heights = True
        if spawn is not None:                    # non-reference diagnostic switch (go1_gym/utils/terrain.py add_terrain_to_map)
            t.origin_height_source = spawn
        if heights_above_terrain:                # non-reference: include/go1sim.h reward_heights_above_terrain
            cfg.rewards.heights_above_terrain = True
        cfg.env.observe_heights = True
        cfg.env.num_observations = cfg.env.num_scalar_observations = 70 + 187
    env = VelocityTrackingEasyEnv(sim_device=f"cuda:{torch.cuda.current_device()}", headless=True, cfg=cfg)
    return HistoryWrapper(env), cfg


def effective_cpus():
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU boxes report 256
    logical CPUs but run the container under a 16-CPU quota; 256 OpenMP threads on that only thrash)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())              # cgroup v1
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(num_envs, target_seconds=15.0):
    """Oracle (fp64 port of the same step, OpenMP over envs) on the host cores: bounded sample of ~target_seconds."""
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    cores = effective_cpus()
    os.environ["OMP_NUM_THREADS"] = str(cores)          # libgomp reads it at load time ...
    torch.set_num_threads(cores)                        # ... and torch has usually loaded it already: set the shared ICV
    import numpy as np
    import pyoracle
    import go1sim_host as H
    from go1_gym.envs.base.legged_robot_config import make_cfg
    from scripts.train_config import apply_train_config
    cfg = apply_train_config(make_cfg(), num_envs=num_envs)
    S, meta = H.build_sim_config(cfg, seed=0)
    B = H.SimBuffers(S, meta, "cpu")
    B.friction_coeffs.uniform_(0.1, 3.0)
    B.payloads.uniform_(-1.0, 3.0)
    orc = pyoracle.Oracle(S, B)
    orc.reset_idx()
    rng = np.random.default_rng(0)
    acts = rng.standard_normal((24, num_envs, 12)).astype(np.float32)
    for a in acts[:8]:              # thread pool spin-up, first-touch page faults
        orc.step(a)
    t0 = time.perf_counter()
    for a in acts[8:16]:
        orc.step(a)
    per_step = (time.perf_counter() - t0) / 8
    policy_steps = int(min(max(24, target_seconds / per_step), 20000))
    t0 = time.perf_counter()
    for i in range(policy_steps):
        orc.step(acts[i % 24])
    dt = time.perf_counter() - t0
    return {"value": num_envs * policy_steps / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{num_envs} envs x {policy_steps} policy steps of the sim step only (no policy/PPO), fp64 oracle "
                      f"(oracle/go1_oracle.c, OpenMP over envs, {cores} threads = cgroup CPU quota), N(0,1) actions, {dt:.1f} s"}


def measure_step_kernel_traffic(envs, timeout=150):
    """HBM bytes per launch of the step kernel from THIS box's counters: two rocprofv3 --pmc passes (FETCH_SIZE, then WRITE_SIZE: they do
    not fit one pass, MI355X_MICROARCH.md) over a short `bench.py --sim-only` child; units KB, factor 1.00 for this kernel's 64-byte SoA
    segments (calibration in profiles/r03_step_kernel_pmc.json).  Returns (bytes, description) or (None, reason): the caller falls back
    to the committed passes."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="go1_pmc_")
        cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
               "--sim-only", "--steps", "2", "--warmup", "1", "--envs", str(envs), "--no-cpu-baseline", "--no-traffic"]
        try:
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout, check=True)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            rows = [r for f in files for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("go1_step_kernel") and r["Counter_Name"] == ctr]
            if not rows:
                return None, f"no {ctr} rows for go1_step_kernel"
            vals[ctr] = sum(float(r["Counter_Value"]) for r in rows) / len(rows)
        except Exception as err:
            return None, f"{ctr} pass failed: {type(err).__name__}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return int(round((vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024)), (f"measured by this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over `bench.py --sim-only` "
                                                                       f"({vals['FETCH_SIZE']:.0f} + {vals['WRITE_SIZE']:.0f} KB per launch)")


def time_sim_only(env, sim, envs, policy_steps, device, warmup=48):
    """SURVEY 8(d) metric 1: env.step alone with pre-generated N(0,1) actions; returns (env-steps/s, mean launch ms)."""
    acts = torch.randn(24, envs, 12, device=device)
    for t in range(warmup):
        env.step(acts[t % 24])
    sim.enable_timing(policy_steps)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(policy_steps):
        env.step(acts[t % 24])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms = sim.read_timings()
    return envs * policy_steps / dt, sum(ms) / max(len(ms), 1)


def time_iterations(runner, env, obs_dict, iters, rollout_only=False, warmup=2):
    T, n = runner.num_steps_per_env, env.num_train_envs

    def one(obs_dict):
        with torch.inference_mode():
            for _ in range(T):
                obs_dict, _ = runner._rollout_step(obs_dict)
            runner.alg.compute_returns(obs_dict["obs_history"][:n], obs_dict["privileged_obs"][:n])
        if rollout_only:
            runner.alg.storage.clear()
        else:
            runner.alg.update()
        return obs_dict
    for _ in range(warmup):
        obs_dict = one(obs_dict)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        obs_dict = one(obs_dict)
    torch.cuda.synchronize()
    return env.num_envs * T * iters / (time.perf_counter() - t0), obs_dict


def listed_contacts(env, steps=12):
    """mean number of contacts the solver lists per environment and substep (top-surface / wall / hip points, self contacts): the `_sig` twin of
    the step kernel stepped over a COPY of the environment's state under N(0, 1) actions (include/go1sim.h GO1_SIG_WORDS) — the contact load the
    launch times of a leg were measured under.  The environment itself is not touched."""
    import go1sim_abi as abi
    import go1sim_host as H
    inner = env.env
    Bt = inner.buffers.clone_to(inner.buffers.device)
    Bt.enable_contact_signature()
    sim = H.Go1Sim(inner.sim_config, Bt, inner.sim_device_id)
    if inner.sim_config_eval is not None:
        sim.set_eval_config(inner.sim_config_eval, inner.num_train_envs)
    sim.set_counters(*inner.sim.counters())
    n = Bt.root_states.shape[1]

    def pop(t):
        t = t.to(torch.int64) & 0xFFFFFFFF
        return sum(((t >> b) & 1) for b in range(32)).float().mean().item()
    acc = {"top": 0.0, "wall": 0.0, "hip": 0.0, "self": 0.0}
    for _ in range(steps):
        sim.step(torch.randn(n, 12, device=Bt.device))
        sig = Bt.contact_signature.view(abi.GO1_SIG_MAX_SUBSTEPS, abi.GO1_SIG_WORDS, n)
        w2 = sig[:, 2].to(torch.int64) & 0xFFFFFFFF
        acc["top"] += pop(sig[:, 0])
        acc["wall"] += pop(sig[:, 1] & 0x1FFF)
        acc["hip"] += pop(sig[:, 1] & ~0x1FFF)
        acc["self"] += sum((((w2 >> (3 * p)) & 7) != 0).float().mean().item() for p in range(6)) + pop((w2 >> 18) & 0xF)
    out = {k: v / steps for k, v in acc.items()}
    out["listed"] = sum(out.values())
    return out


MFMA_PEAK_BF16_TFLOPS = 2500.0            # MI355X_MICROARCH.md: dense bf16 MFMA peak (the 2:1-sparsity headline figure is never used)


def update_roofline(runner, env, obs_dict, reps=3):
    """`PPO.update()` alone (HIP events on the stream it runs on; the rollout that fills the storage is outside the events) against the MFMA
    roofline.  Counted: the first-layer products only — per mini-batch step forward + weight gradient of the PPO pass (n1 rows of W1) and of
    the adaptation pass (its nd rows), 2 M n K each, no input gradient (the history has none) —, which is > 97 % of the update's flops at
    BASELINE configs[2]'s 7744-wide augmented history."""
    from go1_gym_learn.ppo_cse.ppo import PPO_Args
    alg, T, n = runner.alg, runner.num_steps_per_env, env.num_train_envs
    pol = getattr(alg, "policy", None)
    if pol is None or not getattr(alg, "fused", False):
        return None
    ms = []
    for _ in range(reps):
        with torch.inference_mode():
            for _ in range(T):
                obs_dict, _ = runner._rollout_step(obs_dict)
            alg.compute_returns(obs_dict["obs_history"][:n], obs_dict["privileged_obs"][:n])
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        alg.update()
        b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    upd_ms = sum(ms[1:]) / max(len(ms) - 1, 1)               # (the first repetition may re-capture graphs)
    nd, na, nc = pol.first
    mb = n * T // PPO_Args.num_mini_batches
    steps = PPO_Args.num_learning_epochs * PPO_Args.num_mini_batches
    flops = steps * 2.0 * mb * pol.Kp * (2 * (nd + na + nc) + 2 * nd)
    ach = flops / (upd_ms * 1e-3) / 1e12
    return {"bound": "mfma", "kernel": "first-layer GEMMs of the update (hipBLASLt forward + weight gradient, PPO and adaptation pass)",
            "achieved": ach, "peak": MFMA_PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_BF16_TFLOPS, "traffic": None,
            "update_ms": upd_ms, "flops_per_update": flops, "floor_ms_at_peak": flops / (MFMA_PEAK_BF16_TFLOPS * 1e12) * 1e3,
            "note": "achieved = first-layer flops of one update / the WHOLE update's time (optimiser, tails, loss included): a lower bound of the GEMMs' own rate"}


def dropin_default(args, device, iters=40):
    """What the UNCHANGED scripts/train.py gets (train.py:207-216: Runner with the default PPO_Args, `runner.learn`): the caller sets
    nothing on PPO_Args — the bf16 policy is selected by the environment variable GO1_POLICY_DTYPE=bf16 alone —
    and the clock runs over `Runner.learn` itself: logging every iteration, checkpoint + TorchScript export at the end."""
    import tempfile
    from go1_gym_learn.ppo_cse import Runner, RunnerArgs
    from go1_gym_learn.ppo_cse.ppo import PPO_Args
    from ml_logger import logger
    keep = (PPO_Args.autocast_bf16, os.environ.get("GO1_POLICY_DTYPE"), os.getcwd(), RunnerArgs.save_video_interval)
    tmp = tempfile.mkdtemp(prefix="go1_dropin_")
    try:
        PPO_Args.autocast_bf16 = False                          # the class default: what train.py leaves it at
        os.environ["GO1_POLICY_DTYPE"] = "bf16"
        RunnerArgs.save_video_interval = 0
        logger.configure("bench_dropin", root=tmp)
        logger.print_summary = False
        os.chdir(tmp)
        env, _ = build_env(args.envs, 0, args.seed)
        runner = Runner(env, device=device)
        runner.learn(num_learning_iterations=3, init_at_random_ep_len=True, eval_freq=100)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        runner.learn(num_learning_iterations=iters, init_at_random_ep_len=False, eval_freq=100)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return {"env_steps_s": iters * runner.num_steps_per_env * env.num_envs / dt, "iterations": iters, "ms_per_iteration": 1e3 * dt / iters,
                "how": "default PPO_Args + GO1_POLICY_DTYPE=bf16 in the environment, Runner.learn() timed as a whole (per-iteration "
                       "logging, final checkpoint + TorchScript export included)"}
    finally:
        PPO_Args.autocast_bf16 = keep[0]
        if keep[1] is None:
            os.environ.pop("GO1_POLICY_DTYPE", None)
        else:
            os.environ["GO1_POLICY_DTYPE"] = keep[1]
        os.chdir(keep[2])
        RunnerArgs.save_video_interval = keep[3]


def extra_records(args, env, runner, obs_dict, device):
    """The other measurements SURVEY 8(d) defines, in the same JSON line (rank 0, one GPU): the sim-only and
    sim + inference rates of the headline configuration, BASELINE configs[2] (rough terrain + height scan) and the
    per-GPU size of configs[4] (8192 envs).  Each leg is independent: a failure is recorded, never raised."""
    from go1_gym_learn.ppo_cse import Runner
    out = {}
    sim = env.env.sim
    try:
        roll, obs_dict = time_iterations(runner, env, obs_dict, 10, rollout_only=True)
        so, ms = time_sim_only(env, sim, args.envs, 240, device)
        out["rates"] = {"sim_only": so, "rollout_only": roll, "unit": "env-steps/s", "step_kernel_launch_ms_sim_only": ms,
                        "note": "SURVEY 8(d) metrics 1 and 2 on the headline configuration; metric 3 is `value`"}
    except Exception as err:
        out["rates"] = {"error": f"{type(err).__name__}: {err}"}
    try:
        out["update_roofline"] = update_roofline(runner, env, obs_dict)
    except Exception as err:
        out["update_roofline"] = {"error": f"{type(err).__name__}: {err}"}
    try:
        out["dropin_default"] = dropin_default(args, device)
    except Exception as err:
        out["dropin_default"] = {"error": f"{type(err).__name__}: {err}"}
    try:
        from torch.cuda import tunable
        if tunable.is_enabled() and not os.environ.get("GO1_TUNE_ALL"):
            tunable.tuning_enable(False)         # shapes missing from the shipped table run on hipBLASLt's default pick (GO1_TUNE_ALL=1: tools/tune_gemms.sh)
        env3, _ = build_env(args.envs, 0, args.seed, rough=True)
        runner3 = Runner(env3, device=device)
        env3.episode_length_buf.copy_(torch.randint_like(env3.episode_length_buf, high=int(env3.max_episode_length)))
        od3 = env3.get_observations()
        full, od3 = time_iterations(runner3, env3, od3, 5, warmup=2)
        roof3 = update_roofline(runner3, env3, od3)
        so, ms = time_sim_only(env3, env3.env.sim, args.envs, 240, device)
        f3 = env3.env.extras["sim_faults"].consume()
        load3 = listed_contacts(env3)
        # the same kernel on a FRESH environment of the same configuration, in this process (reset distribution, no policy iterations before):
        # the regime of `bench.py --sim-only --rough` (profiles/r06_walls_regimes.txt reconciles the two)
        env3b, _ = build_env(args.envs, 0, args.seed, rough=True)
        env3b.episode_length_buf.copy_(torch.randint_like(env3b.episode_length_buf, high=int(env3b.max_episode_length)))
        so_fresh, ms_fresh = time_sim_only(env3b, env3b.env.sim, args.envs, 240, device)
        load3b = listed_contacts(env3b)
        del env3b
        out["rough_trimesh"] = {"workload": "BASELINE configs[2]: terrain-curriculum tile grid (slopes, rough slopes, stairs up / down, discrete "
                                            "obstacles) as a `trimesh` terrain — int16 height field with vertical faces where the slope exceeds "
                                            "slope_treshold 0.75 —, 187-point height scan in the observation (257 wide, history 7710), 4096 envs",
                                "sim_ppo_env_steps_s": full, "sim_only_env_steps_s": so, "step_kernel_launch_ms": ms,
                                "wall_instance": bool(env3.env.sim_config.hf_wall_units > 0),
                                "listed_contacts_per_env_and_substep": load3,
                                "fresh_env_same_process": {"sim_only_env_steps_s": so_fresh, "step_kernel_launch_ms": ms_fresh,
                                                           "listed_contacts_per_env_and_substep": load3b},
                                "roofline": roof3,
                                "guard_activations": {k: v for k, v in f3.items() if v},
                                "note": "sim+PPO: 5 timed PPO iterations (live policy), first-layer GEMM selections for the 7744-wide history from the "
                                        "shipped TunableOp table (1.17 TFLOP of first-layer products per mini-batch step: the update is FLOP-bound, "
                                        "DESIGN.md section 9); sim only: N(0,1) actions"}
        del runner3, env3
    except Exception as err:
        out["rough_trimesh"] = {"error": f"{type(err).__name__}: {err}"}
    try:
        from torch.cuda import tunable
        if tunable.is_enabled() and not os.environ.get("GO1_TUNE_ALL"):
            tunable.tuning_enable(False)         # shapes missing from the shipped table run on hipBLASLt's default pick
        env8, _ = build_env(8192, 0, args.seed)
        runner8 = Runner(env8, device=device)
        env8.episode_length_buf.copy_(torch.randint_like(env8.episode_length_buf, high=int(env8.max_episode_length)))
        od8 = env8.get_observations()
        full, od8 = time_iterations(runner8, env8, od8, 5, warmup=3)
        so, ms = time_sim_only(env8, env8.env.sim, 8192, 240, device)
        out["envs_8192"] = {"workload": "per-GPU size of BASELINE configs[4]: 8192 envs, train.py configuration, one GPU; first-layer GEMM "
                                        "selections from the shipped TunableOp table", "env_steps_s": full, "sim_only_env_steps_s": so,
                            "step_kernel_launch_ms": ms}
        del runner8, env8
    except Exception as err:
        out["envs_8192"] = {"error": f"{type(err).__name__}: {err}"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10, help="timed PPO iterations")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--envs", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--fp32", action="store_true", help="fp32 policy instead of bf16 autocast")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sim-only", action="store_true", help="diagnostic: time env.step alone with pre-generated actions")
    ap.add_argument("--zero-actions", action="store_true", help="diagnostic with --sim-only: standing robots (few resets)")
    ap.add_argument("--rollout-only", action="store_true", help="diagnostic: sim + policy inference + storage, no update (SURVEY 8d metric 2)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for single-GPU dry runs)")
    ap.add_argument("--same-device", action="store_true", help="dry run: all ranks share GPU 0 (with --backend gloo)")
    ap.add_argument("--grad-dtype", default="fp32", choices=["fp32", "bf16"], help="N > 1: dtype of the PPO-stage gradient exchange (PPO_Args.dp_grad_dtype)")
    ap.add_argument("--zero1", action="store_true", help="N > 1: reduce-scatter + sharded optimiser step + all-gather (PPO_Args.dp_zero1)")
    ap.add_argument("--rough", action="store_true", help="diagnostic: BASELINE configs[2] (trimesh tile grid + height scan) instead of the flat headline "
                    "configuration — with --sim-only for profiles of go1_step_kernel_walls; never the headline line")
    ap.add_argument("--breakdown", action="store_true", help="diagnostic: print rollout/update split to stderr (adds syncs)")
    ap.add_argument("--headline-only", action="store_true", help="skip the extra single-GPU records (rates, height field, 8192 envs)")
    ap.add_argument("--no-traffic", action="store_true", help="do not run the two rocprofv3 --pmc passes that measure the step kernel's HBM traffic "
                                                              "(roofline.traffic then comes from the committed passes in profiles/)")
    ap.add_argument("--curriculum-interval", type=int, default=1,
                    help="commands.curriculum_update_interval K: 1 = the reference's per-step curriculum update (curriculum.py) at EVERY rank count, so "
                         "that the 1-GPU headline and the N-GPU scaling numbers run the same algorithm; K > 1 coalesces the sharded run's "
                         "success-count all-reduce to one per K steps (samples inside the window see the weights from its start)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    force_dp = os.environ.get("GO1_FORCE_DP", "0") not in ("", "0")        # one-rank process group: the DP path on a 1-GPU box
    use_dist = world > 1 or force_dp
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import datetime
        tmo = datetime.timedelta(seconds=600)          # ranks may start minutes apart (first `import torch` on a fresh box); a dead peer must not hang the job
        if args.backend == "gloo":
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # single-node dry runs: do not depend on the host name resolving
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=tmo)
        else:
            dist.init_process_group(args.backend, timeout=tmo)

    dp_trace = None
    if os.environ.get("GO1_DP_TRACE", "0") not in ("", "0") and use_dist:
        # count what actually executes (tests/test_gpu_env.py, the one-rank RCCL run): collectives by kind, graph replays
        dp_trace = {"backend": dist.get_backend(), "all_reduce": 0, "curriculum_all_reduce": 0, "reduce_scatter": 0, "all_gather": 0,
                    "broadcast": 0, "graph_replays": 0}

        def _counted(name, fn):
            def wrapped(t, *a, **k):
                key = "curriculum_all_reduce" if name == "all_reduce" and t.dtype == torch.int32 else name
                dp_trace[key] += 1
                return fn(t, *a, **k)
            return wrapped
        dist.all_reduce = _counted("all_reduce", dist.all_reduce)
        dist.reduce_scatter_tensor = _counted("reduce_scatter", dist.reduce_scatter_tensor)
        dist.all_gather_into_tensor = _counted("all_gather", dist.all_gather_into_tensor)
        dist.broadcast = _counted("broadcast", dist.broadcast)
        _replay = torch.cuda.CUDAGraph.replay

        def replay(self):
            dp_trace["graph_replays"] += 1
            return _replay(self)
        torch.cuda.CUDAGraph.replay = replay

    from go1_gym_learn.ppo_cse import Runner, RunnerArgs
    from go1_gym_learn.ppo_cse.ppo import PPO_Args
    PPO_Args.autocast_bf16 = not args.fp32
    env_dtype = os.environ.get("GO1_POLICY_DTYPE", "").strip().lower()
    if env_dtype and (env_dtype in ("bf16", "bfloat16")) == bool(args.fp32):      # the variable would override what this line is labelled with
        raise SystemExit(f"GO1_POLICY_DTYPE={env_dtype} contradicts {'--fp32' if args.fp32 else 'the bf16 default'}: unset it")
    PPO_Args.dp_grad_dtype, PPO_Args.dp_zero1 = args.grad_dtype, bool(args.zero1)
    RunnerArgs.save_video_interval = 0
    torch.manual_seed(args.seed + rank)
    # the command curriculum keeps the reference's per-step cadence at every rank count (sharded: one 7 KB int32 all-reduce of the
    # success counts per step); --curriculum-interval 24 exchanges them once per rollout instead — a different sampling cadence,
    # recorded in config.curriculum_update_interval
    env, cfg = build_env(args.envs, rank, args.seed, rough=args.rough, curriculum_update_interval=args.curriculum_interval)
    device = f"cuda:{local_rank}"
    runner = Runner(env, device=device)
    sim = env.env.sim
    T = runner.num_steps_per_env
    sim.enable_timing(T * args.steps)

    buf = env.episode_length_buf
    buf.copy_(torch.randint_like(buf, high=int(env.max_episode_length)))
    obs_dict = env.get_observations()
    runner.alg.actor_critic.train()

    split = {"rollout": 0.0, "update": 0.0}

    def iteration(obs_dict):
        if args.breakdown:
            torch.cuda.synchronize()
            ta = time.perf_counter()
        with torch.inference_mode():
            for _ in range(T):
                obs_dict, _ = runner._rollout_step(obs_dict)
            n = env.num_train_envs
            runner.alg.compute_returns(obs_dict["obs_history"][:n], obs_dict["privileged_obs"][:n])
        if args.breakdown:
            torch.cuda.synchronize()
            tb = time.perf_counter()
        if args.rollout_only:
            runner.alg.storage.clear()
            return obs_dict
        runner.alg.update()
        if args.breakdown:
            torch.cuda.synchronize()
            split["rollout"] += tb - ta
            split["update"] += time.perf_counter() - tb
        return obs_dict

    def sim_iteration(obs_dict, acts):
        for t in range(T):
            obs_dict, _, _, _ = env.step(acts[t])
        return obs_dict

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    acts = (torch.zeros if args.zero_actions else torch.randn)(T, args.envs, 12, device=device) if args.sim_only else None
    for _ in range(args.warmup):
        obs_dict = sim_iteration(obs_dict, acts) if args.sim_only else iteration(obs_dict)
    barrier()
    split["rollout"] = split["update"] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        obs_dict = sim_iteration(obs_dict, acts) if args.sim_only else iteration(obs_dict)
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)

    kernel_ms = sim.read_timings()
    if dp_trace is not None and rank == 0:
        dp_trace.update(dp=bool(runner.alg.dp), curriculum_sync=bool(env.env._curriculum_sync), world=world)
        print("[dp-trace]" + json.dumps(dp_trace), file=sys.stderr)
    if args.breakdown and rank == 0:
        n_it = args.steps
        print(f"[breakdown] rollout {1e3 * split['rollout'] / n_it:.1f} ms/iter, update {1e3 * split['update'] / n_it:.1f} ms/iter", file=sys.stderr)
        try:
            from torch.cuda import tunable
            print(f"[breakdown] TunableOp enabled={tunable.is_enabled()} table entries={len(tunable.get_results())} file={tunable.get_filename()}",
                  file=sys.stderr)
        except Exception as err:
            print(f"[breakdown] TunableOp state unavailable: {err}", file=sys.stderr)
    if rank == 0:
        total_env_steps = args.envs * T * args.steps * world
        avg_ms = sum(kernel_ms) / max(len(kernel_ms), 1)
        valu_insts = None
        traffic, traffic_src = None, None   # HBM bytes per launch: NOT measured by this run — read from the committed PMC passes
        for name in ("r05_step_kernel_pmc.json", "r04_step_kernel_pmc.json", "r03_step_kernel_pmc.json", "r02_step_kernel_pmc.json", "r01_step_kernel_pmc.json"):
            try:
                with open(os.path.join(REPO, "profiles", name)) as f:
                    pmc = json.load(f)
                if args.envs == 4096:
                    traffic = pmc["hbm_traffic_bytes_per_launch"]
                    traffic_src = f"profiles/{name} (committed rocprofv3 --pmc passes of this kernel at 4096 envs, not this run)"
                    valu_insts = pmc.get("per_launch", {}).get("SQ_INSTS_VALU")
                break
            except (OSError, KeyError, ValueError):
                continue
        if world == 1 and not args.no_traffic and not args.headline_only and not (args.sim_only or args.rollout_only) and args.envs == 4096:
            measured, how = measure_step_kernel_traffic(args.envs)
            if measured is not None:
                traffic, traffic_src = measured, how
            else:
                traffic_src = f"{traffic_src}; live measurement unavailable ({how})"
        # what the kernel IS bound by: vector-ALU issue of the master wavefronts.  Wave-level VALU instructions (committed PMC pass)
        # x 64 lanes over this run's launch time, against the fp32 vector rate (157.3 TFLOP/s = 78.6 T lane-FMAs/s,
        # MI355X_MICROARCH.md) — an issue-slot fraction, not a FLOP count (moves, compares and selects occupy slots too)
        valu = None
        if valu_insts and avg_ms > 0:
            lane_ops = valu_insts * 64 / (avg_ms * 1e-3)
            valu = {"wave_instructions_per_launch": valu_insts, "lane_ops_per_s": lane_ops, "peak_lane_ops_per_s": 78.65e12,
                    "frac_of_fp32_vector_issue_peak": lane_ops / 78.65e12,
                    "note": "chip-wide; the step's serial spine runs on 256 of the 1024 SIMDs (one master wavefront per CU), the helper "
                            "wavefronts fill the other SIMDs only during the actuator network and the terrain-contact row emission"}
        faults = env.env.extras["sim_faults"].consume()
        bytes_per_launch = ALGORITHMIC_BYTES_PER_ENV_STEP * args.envs
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        out = {
            "metric": ("env-steps/sec (sim only, diagnostic)" if args.sim_only else
                       "env-steps/sec (sim + policy inference, diagnostic)" if args.rollout_only else "env-steps/sec (sim+PPO)"),
            "value": total_env_steps / elapsed, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("DIAGNOSTIC (--rough), not the headline: BASELINE configs[2], trimesh tile grid + 187-point height scan, "
                                    "train.py config otherwise") if args.rough else
                                   "Go1 flat terrain, 4096 envs/GPU, train.py config (actuator_net, lag 6, DR, gait "
                                   "curriculum), HIP sim + ppo_cse, 24 steps/iter, 5 epochs x 4 minibatches",
                       "envs_per_gpu": args.envs, "curriculum_update_interval": int(args.curriculum_interval), "policy_dtype": "fp32" if args.fp32 else "bf16 autocast (fp32 master)",
                       "physics_dtype": "f32", "step": "one PPO iteration = 24 x envs env-steps + update",
                       "parallelism": (f"dp{world} (envs sharded, {args.backend} gradient "
                                       f"{'reduce-scatter + sharded step + all-gather' if args.zero1 else 'all-reduce'}, {args.grad_dtype})")
                       if use_dist else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": "go1_step_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "launch_ms": avg_ms, "launches": len(kernel_ms), "algorithmic_bytes_per_launch": bytes_per_launch,
                         "valu": valu,
                         "note": "latency/issue-bound O(n_dof) recursion: the HBM fraction is reported as north_star requires, "
                                 "it is not the limiter (DESIGN.md Measurement)"},
        }
        out["guard_activations"] = {"fatal": faults["fatal"], "by_site": {k: v for k, v in faults.items() if k != "fatal" and v},
                                    "env_steps": args.envs * T * (args.steps + args.warmup),
                                    "note": "containments of failed environments (include/go1sim.h Go1FaultBit) during warm-up + timed steps"}
        if world == 1 and not args.headline_only and not (args.sim_only or args.rollout_only):
            # the driver times --steps 20 (0.45 s): the same measurement over a 200-iteration window next to it
            long_iters = 200
            rate, obs_dict = time_iterations(runner, env, obs_dict, long_iters, warmup=0)
            out["long_window"] = {"iterations": long_iters, "value": rate, "ms_per_step": 1e3 * args.envs * T / rate, "unit": "env-steps/s",
                                  "note": "same workload and code path as `value`, timed over 200 PPO iterations in one window"}
            out.update(extra_records(args, env, runner, obs_dict, device))
        try:
            with open(os.path.join(REPO, "profiles", "reference_python_maps_cpu.json")) as f:
                out["reference_python_maps_cpu"] = json.load(f)       # produced where /root/reference exists (tools/time_reference_maps.py)
        except (OSError, ValueError):
            out["reference_python_maps_cpu"] = None
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.envs)
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
