"""Data-parallel PPO path on CPU: 2 processes, gloo backend (the GPU path is the same code over RCCL).
Environments are partitioned per rank, the policy is replicated: after an update every rank must hold the same
weights and learning rate, advantage statistics must be global, and the all-reduced gradient must be the mean of
the per-rank gradients."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, out, grad_dtype="fp32", zero1=False):
    sys.path[:0] = [os.path.join(HERE, "..", "walk-these-ways_amd", "shims"), os.path.join(HERE, "..", "walk-these-ways_amd")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")          # (gloo otherwise resolves the host name, which may not resolve in a container)
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    from go1_gym_learn.ppo_cse.actor_critic import AC_Args, ActorCritic
    from go1_gym_learn.ppo_cse.ppo import PPO, PPO_Args
    from go1_gym_learn.ppo_cse.rollout_storage import RolloutStorage
    AC_Args.actor_hidden_dims, AC_Args.critic_hidden_dims, AC_Args.adaptation_module_branch_hidden_dims = [32, 16], [24, 16], [16, 8]
    PPO_Args.dp_grad_dtype, PPO_Args.dp_zero1 = grad_dtype, zero1
    N, T, no, npv, H, na = 12, 5, 10, 2, 3, 12
    torch.manual_seed(100 + rank)                     # different initial weights: rank 0's must win
    alg = PPO(ActorCritic(no, npv, no * H, na), device="cpu")
    w0 = alg.flat_param.clone()
    alg.init_storage(N, T, [no], [npv], [no * H], [na])
    g = torch.Generator().manual_seed(7 + rank)       # different data per rank (env shard)
    for t in range(T):
        obs, priv, hist = torch.randn(N, no, generator=g), torch.randn(N, npv, generator=g), torch.randn(N, no * H, generator=g)
        alg.act(obs, priv, hist)
        alg.process_env_step(torch.randn(N, generator=g), (torch.rand(N, generator=g) < 0.1).to(torch.uint8),
                             {"env_bins": torch.zeros(N), "time_outs": torch.zeros(N, dtype=torch.bool)})
    alg.compute_returns(hist, priv)
    adv_local = (alg.storage.returns - alg.storage.values).clone()
    gathered = [torch.zeros_like(adv_local) for _ in range(world)]
    dist.all_gather(gathered, adv_local)
    allv = torch.cat(gathered)
    expect = (adv_local - allv.mean()) / (allv.std() + 1e-8)
    adv_ok = torch.allclose(alg.storage.advantages, expect, atol=1e-5)
    # gradient all-reduce == mean of per-rank gradients (adaptation stage: all-reduce, then divide by world size)
    alg.master.grad.copy_(torch.full_like(alg.master.grad, float(rank + 1)))
    dist.all_reduce(alg.master.grad)
    lr_save = alg.adaptation_module_optimizer.param_groups[0]["lr"]
    alg.adaptation_module_optimizer.param_groups[0]["lr"] = 0.0
    alg._stage_adapt_step()
    alg.adaptation_module_optimizer.param_groups[0]["lr"] = lr_save
    alg.adaptation_module_optimizer.state.clear()
    grad_ok = torch.allclose(alg.master.grad, torch.full_like(alg.master.grad, (1 + world) / 2))
    torch.manual_seed(5)
    alg.update()
    out[rank] = dict(w0=w0, w=alg.flat_param.clone(), lr=alg.learning_rate, adv_ok=adv_ok, grad_ok=grad_ok, dp=alg.dp)
    dist.destroy_process_group()


_DP_RESULTS = {}


@pytest.mark.parametrize("grad_dtype,zero1,world", [("fp32", False, 2), ("bf16", False, 2), ("fp32", True, 2), ("bf16", True, 2),
                                                    ("fp32", True, 3)])
def test_two_rank_gloo_update_keeps_replicas_identical(grad_dtype, zero1, world):
    """every exchange mode (PPO_Args.dp_grad_dtype, PPO_Args.dp_zero1): both ranks end with bit-identical weights and
    learning rate; the sharded step (reduce-scatter, every rank steps its slice with the global norm / KL, all-gather)
    reproduces the all-reduce step's weights to round-off, the bf16 exchange stays within bf16 gradient noise of it.  Three
    ranks: the parameter count is not a multiple of the world size (the flat master is padded for the scatter)."""
    port = 29500 + (os.getpid() + 17 * len(_DP_RESULTS)) % 2000
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out, grad_dtype, zero1), nprocs=world, join=True)
    r0, r1 = out[0], out[world - 1]
    assert all(torch.equal(out[r]["w"], r0["w"]) for r in range(world))
    if world == 2:
        _DP_RESULTS[(grad_dtype, zero1)] = r0["w"].clone()
    if world == 2 and (grad_dtype, zero1) != ("fp32", False) and ("fp32", False) in _DP_RESULTS:
        base = _DP_RESULTS[("fp32", False)]
        n = base.numel()
        tol = 1e-5 if grad_dtype == "fp32" else 5e-3           # Adam's normalised step amplifies the bf16 rounding of small gradients
        assert float((r0["w"][:n] - base).abs().max()) < tol, (grad_dtype, zero1, float((r0["w"][:n] - base).abs().max()))
    assert r0["dp"] and r1["dp"]
    assert torch.equal(r0["w0"], r1["w0"])                 # broadcast of rank 0's initial weights
    assert torch.equal(r0["w"], r1["w"])                   # identical after 20 + 20 optimiser steps
    assert not torch.equal(r0["w"], r0["w0"])
    assert r0["lr"] == r1["lr"]
    assert r0["adv_ok"] and r1["adv_ok"]
    assert r0["grad_ok"] and r1["grad_ok"]


# ---- one global command curriculum over sharded environments ---------------------------------------------
def _curriculum_run(rank, world, n_local, steps, interval=1):
    """LeggedRobot.step on the oracle-backed stand-in (tests/fake_sim.py): `world` shards of `n_local` envs, or one run
    over all of them.  Per-env inputs are functions of the GLOBAL env id, so only the sharding differs."""
    import types
    import fake_sim
    import go1sim_host as H
    from go1_gym.envs.base import base_task
    from go1_gym.envs.base.legged_robot_config import make_cfg
    from go1_gym.envs.go1.velocity_tracking import VelocityTrackingEasyEnv
    from scripts.train_config import apply_train_config
    base_task.BaseTask._resolve_device = lambda self, d: (setattr(self, "sim_device_id", 0), "cpu")[1]
    H.Go1Sim = fake_sim.OracleBackedSim
    cfg = apply_train_config(make_cfg(), num_envs=n_local)
    cfg.env.env_id_offset = rank * n_local
    cfg.terrain.mesh_type = "plane"
    cfg.commands.resampling_time = 0.16                                  # 8 steps: many interval resamples
    cfg.commands.curriculum_update_interval = interval                   # steps between weight updates (= between exchanges when sharded)
    for k in ("tracking_lin_vel", "tracking_ang_vel", "tracking_contacts_shaped_force", "tracking_contacts_shaped_vel"):
        setattr(cfg.curriculum_thresholds, k, 0.05)                      # successes do happen under random actions
    torch.manual_seed(0)
    env = VelocityTrackingEasyEnv(sim_device="cuda:0", headless=True, cfg=cfg)
    gid = torch.arange(n_local) + rank * n_local
    B = env.buffers
    B.payloads[:] = -1.0 + 4.0 * ((gid * 37) % 101) / 101.0
    B.friction_coeffs[:] = 0.1 + 2.9 * ((gid * 53) % 89) / 89.0
    B.restitutions[:] = 0.4 * ((gid * 29) % 97) / 97.0
    B.com_displacements.zero_()
    B.env_origins.zero_()
    env.reset()
    g = torch.Generator().manual_seed(3)
    acts = 0.4 * torch.randn(steps, n_local * world if world > 1 else n_local, 12, generator=g)
    lo = rank * n_local if world > 1 else 0
    for t in range(steps):
        env.step(acts[t, lo:lo + n_local].contiguous())
    return dict(weights=B.curriculum_weights.clone(), commands=B.commands.clone(), bins=B.env_command_bins.clone(),
                sync=env._curriculum_sync)


def _curriculum_worker(rank, world, port, out, interval=1):
    for p in (os.path.join(HERE, "..", "walk-these-ways_amd", "shims"), os.path.join(HERE, "..", "walk-these-ways_amd"),
              os.path.join(HERE, "..", "oracle"), os.path.join(HERE, ".."), HERE):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out[rank] = _curriculum_run(rank, world, 48, 40, interval)
    dist.destroy_process_group()


@pytest.mark.parametrize("interval", [1, 8])
def test_two_rank_curriculum_equals_single_rank_over_the_concatenated_shards(interval):
    """interval 1: the success counts are exchanged after every step (the reference's update cadence); interval 8: the counts of 8
    steps sit in 8 slots, ONE all-reduce carries them and the 8 per-step updates are applied in order (what bench.py uses over
    ranks, with the rollout length) — either way the sharded run IS the single-process run with the same interval."""
    for p in (os.path.join(HERE, "..", "oracle"), os.path.join(HERE, ".."), HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    world = 2
    port = 31500 + os.getpid() % 2000
    out = mp.Manager().dict()
    mp.spawn(_curriculum_worker, args=(world, port + interval, out, interval), nprocs=world, join=True)
    import go1sim_host as H
    from go1_gym.envs.base import base_task
    saved = (H.Go1Sim, base_task.BaseTask._resolve_device)
    try:
        single = _curriculum_run(0, 1, 96, 40, interval)
    finally:
        H.Go1Sim, base_task.BaseTask._resolve_device = saved
    r0, r1 = out[0], out[1]
    assert r0["sync"] and r1["sync"] and not single["sync"]
    assert torch.equal(r0["weights"], r1["weights"])                               # one curriculum on every rank ...
    assert torch.equal(r0["weights"], single["weights"])                           # ... and it is the single-GPU one
    assert bool(((single["weights"] > 0) & (single["weights"] < 1)).any())           # the frontier moved: successes were counted
    assert torch.equal(torch.cat((r0["commands"], r1["commands"]), dim=1), single["commands"])
    assert torch.equal(torch.cat((r0["bins"], r1["bins"])), single["bins"])
