#!/usr/bin/env python3
"""tests/golden/runner_iteration.npz: the REFERENCE `go1_gym_learn.ppo_cse.Runner` (and its PPO / ActorCritic / RolloutStorage)
driving THIS repository's environment classes for two learning iterations on the CPU.

The environment is the product's `VelocityTrackingEasyEnv` + `HistoryWrapper` over the oracle-backed stand-in simulator of
tests/fake_sim.py (deterministic given the seeds), so the fixture pins the learner's glue — when observations are read, what
the storage receives, bootstrapping, advantage normalisation, mini-batch order, the two optimiser steps — end to end: the
product's Runner must arrive at the same weights (tests/test_dropin_config.py).  It also shows that a user may keep the
reference's own Runner on top of this environment.  Needs /root/reference; run by make_golden.py in its own process (the two
`go1_gym_learn` packages cannot live in one interpreter).

Regeneration policy: the trajectories come from THIS repository's physics contract (oracle/go1_oracle.c through tests/fake_sim.py), so
both fixtures are regenerated whenever that contract changes — last in round 3's physics commit (8893c20: 24-contact matrix-free
solve, trunk corners, thigh capsules, static / dynamic cone), which moved the weights after two iterations by ~1e-4 without any
change to this script.  A change of the learner's glue alone must NOT require regenerating them."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
REPO = os.path.dirname(TESTS)
PKG = os.path.join(REPO, "walk-these-ways_amd")
REF = "/root/reference"

SETTINGS = dict(num_envs=32, num_steps_per_env=6, iterations=2, actor=[32, 16], critic=[24, 16], adaptation=[16, 8], seed=0,
                episode_length_s=0.16, num_eval_envs=0, name="runner_iteration.npz")
# the same with 16 evaluation environments behind the training ones (eval_cfg: the Runner appends the deterministic student
# actions for them, ppo_cse/__init__.py:139-147, and learns from the training ones only)
SETTINGS_EVAL = dict(SETTINGS, num_eval_envs=16, seed=3, name="runner_iteration_eval.npz")


def build_env(settings):
    """(shared with the test) the product's environment on the stand-in simulator"""
    import torch
    from go1_gym.envs.base.legged_robot_config import make_cfg
    from go1_gym.envs.go1.velocity_tracking import VelocityTrackingEasyEnv
    from go1_gym.envs.wrappers.history_wrapper import HistoryWrapper
    from scripts.train_config import apply_train_config
    cfg = apply_train_config(make_cfg(), num_envs=settings["num_envs"])
    cfg.terrain.mesh_type = "plane"
    cfg.env.episode_length_s = settings["episode_length_s"]          # resets and time-outs inside the two iterations
    ev = None
    if settings.get("num_eval_envs", 0):
        ev = apply_train_config(make_cfg(), num_envs=settings["num_eval_envs"])
        ev.terrain.mesh_type = "plane"
        ev.env.episode_length_s = settings["episode_length_s"]
        ev.domain_rand.friction_range = [2.0, 2.5]
    torch.manual_seed(settings["seed"])
    return HistoryWrapper(VelocityTrackingEasyEnv(sim_device="cuda:0", headless=True, cfg=cfg, eval_cfg=ev))


def main(SETTINGS):
    for p in (os.path.join(PKG, "shims"), PKG, os.path.join(REPO, "oracle"), REPO, TESTS):
        sys.path.insert(0, p)
    import numpy as np
    import pytest
    import torch
    import fake_sim
    mp = pytest.MonkeyPatch()
    fake_sim.install(mp)
    env = build_env(SETTINGS)                                        # the product's go1_gym is imported now
    assert not any(m.startswith("go1_gym_learn") for m in sys.modules), [m for m in sys.modules if m.startswith("go1_gym_learn")]
    sys.path.insert(0, REF)                                          # ... and go1_gym_learn resolves to the REFERENCE from here on
    from go1_gym_learn.ppo_cse import Runner, RunnerArgs
    from go1_gym_learn.ppo_cse.actor_critic import AC_Args
    import go1_gym_learn
    assert go1_gym_learn.__file__.startswith(REF), go1_gym_learn.__file__
    from ml_logger import logger
    import tempfile
    tmp = tempfile.mkdtemp()
    logger.configure("runner_iteration", root=tmp)
    logger.print_summary = False
    os.chdir(tmp)
    AC_Args.actor_hidden_dims, AC_Args.critic_hidden_dims = SETTINGS["actor"], SETTINGS["critic"]
    AC_Args.adaptation_module_branch_hidden_dims = SETTINGS["adaptation"]
    RunnerArgs.num_steps_per_env = SETTINGS["num_steps_per_env"]
    RunnerArgs.save_video_interval = 0
    torch.manual_seed(SETTINGS["seed"] + 1)
    runner = Runner(env, device="cpu")
    init = {k: v.clone() for k, v in runner.alg.actor_critic.state_dict().items()}
    runner.learn(num_learning_iterations=SETTINGS["iterations"], init_at_random_ep_len=False, eval_freq=100)
    st = runner.alg.storage
    out = {"init_" + k: v.numpy() for k, v in init.items()}
    out.update({"final_" + k: v.detach().numpy() for k, v in runner.alg.actor_critic.state_dict().items()})
    out.update(last_actions=st.actions.numpy(), last_rewards=st.rewards.numpy(), last_dones=st.dones.numpy(), last_values=st.values.numpy(),
               last_returns=st.returns.numpy(), last_advantages=st.advantages.numpy(), lr=np.array(runner.alg.learning_rate),
               tot_timesteps=np.array(runner.tot_timesteps))
    import json
    np.savez_compressed(os.path.join(HERE, SETTINGS["name"]), settings=np.array(json.dumps(SETTINGS)), **out)
    print(SETTINGS["name"], "dones in the last rollout", int(st.dones.sum()), "lr", runner.alg.learning_rate, "timesteps", runner.tot_timesteps)
    mp.undo()


if __name__ == "__main__":
    main(SETTINGS_EVAL if sys.argv[1:] == ["eval"] else SETTINGS)
