"""Train / evaluation environment split (`eval_cfg`; reference base_task.py:43-49, legged_robot.py:41-44, 181-195, 531-544,
ppo_cse/__init__.py:139-154) through the product's host classes.  No GPU here: the simulator handle is the oracle-backed
stand-in of tests/fake_sim.py on CPU buffers (the kernel side of the split is covered by tests/test_emu_parity.py and
tests/test_gpu_parity.py)."""
import pytest
import torch


def _cfgs(n_train, n_eval):
    from go1_gym.envs.base.legged_robot_config import make_cfg
    from scripts.train_config import apply_train_config
    cfg = apply_train_config(make_cfg(), num_envs=n_train)
    ev = apply_train_config(make_cfg(), num_envs=n_eval)          # (a second config tree: the sections are not deep-copyable)
    for c in (cfg, ev):
        c.terrain.mesh_type = "plane"
        c.env.episode_length_s = 0.4            # 20 policy steps: every environment finishes episodes during the test
    ev.env.env_spacing = 7.0
    ev.domain_rand.friction_range = [5.0, 5.5]
    ev.domain_rand.added_mass_range = [4.0, 4.5]
    ev.domain_rand.motor_strength_range = [1.5, 1.6]
    return cfg, ev


def test_env_with_eval_cfg_and_runner_iteration(monkeypatch, tmp_path):
    import fake_sim
    from go1_gym.envs.go1.velocity_tracking import VelocityTrackingEasyEnv
    from go1_gym.envs.wrappers.history_wrapper import HistoryWrapper
    from go1_gym_learn.ppo_cse import Runner, RunnerArgs
    from ml_logger import logger
    fake_sim.install(monkeypatch)
    NT, NE = 32, 16
    cfg, ev = _cfgs(NT, NE)
    torch.manual_seed(0)
    env = VelocityTrackingEasyEnv(sim_device="cuda:0", headless=True, cfg=cfg, eval_cfg=ev)
    assert (env.num_envs, env.num_train_envs, env.num_eval_envs) == (NT + NE, NT, NE)
    B = env.buffers
    # set-up time draws come from each group's own ranges (legged_robot.py:1548 through _call_train_eval)
    assert bool(((B.friction_coeffs[NT:] >= 5.0) & (B.friction_coeffs[NT:] <= 5.5)).all()) and bool((B.friction_coeffs[:NT] < 5.0).all())
    assert bool(((B.payloads[NT:] >= 4.0) & (B.payloads[NT:] <= 4.5)).all())
    # one origin grid per group, the evaluation one with its own spacing (:1538, :1704-1714)
    assert float(env.env_origins[NT:, 0].max()) == pytest.approx(7.0 * 3) and float(env.env_origins[:NT, 0].max()) < 7.0 * 3
    assert env.extras["env_bins"].shape[0] == NT and env.extras["time_outs"].shape[0] == NT and env.extras["eval/episode"] == {}
    assert bool((B.episode_sums_eval == -1).all())
    wrapped = HistoryWrapper(env)
    logger.configure("evalsplit", root=str(tmp_path))
    logger.print_summary = False
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(RunnerArgs, "save_video_interval", 0)
    monkeypatch.setattr(RunnerArgs, "num_steps_per_env", 24)
    runner = Runner(wrapped, device="cpu")
    assert runner.alg.storage.num_envs == NT                      # only the training environments are learnt from
    seen = []
    real_step = wrapped.env.step.__func__

    def spy(self, actions):
        seen.append(tuple(actions.shape))
        return real_step(self, actions)

    monkeypatch.setattr(type(wrapped.env), "step", spy)
    runner.learn(num_learning_iterations=1, init_at_random_ep_len=True, eval_freq=100)
    assert seen and all(s == (NT + NE, 12) for s in seen)          # train actions + deterministic student actions (:139-147)
    # re-drawn motor strengths (every reset) follow the group's range; the evaluation episodes are remembered, not logged
    ms = B.motor_strengths
    assert bool(((ms[:, NT:] >= 1.5) & (ms[:, NT:] <= 1.6)).all()) and bool((ms[:, :NT] < 1.5).all())
    done = B.episode_sums_eval[-1] != -1
    assert int(done[NT:].sum()) > 0 and int(done[:NT].sum()) == 0
    assert env.episode_sums_eval["total"].data_ptr() == B.episode_sums_eval[-1].data_ptr()


def test_eval_cfg_needs_wavefront_aligned_train_count(monkeypatch):
    import fake_sim
    from go1_gym.envs.go1.velocity_tracking import VelocityTrackingEasyEnv
    fake_sim.install(monkeypatch)
    cfg, ev = _cfgs(24, 8)
    with pytest.raises(ValueError, match="multiple of 16"):
        VelocityTrackingEasyEnv(sim_device="cuda:0", headless=True, cfg=cfg, eval_cfg=ev)
