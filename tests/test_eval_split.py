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
    from util import check_exported_policy_layout
    check_exported_policy_layout(str(tmp_path / "evalsplit" / "checkpoints"))      # (no reference tree needed for this one)


def test_eval_cfg_needs_wavefront_aligned_train_count(monkeypatch):
    import fake_sim
    from go1_gym.envs.go1.velocity_tracking import VelocityTrackingEasyEnv
    fake_sim.install(monkeypatch)
    cfg, ev = _cfgs(24, 8)
    with pytest.raises(ValueError, match="multiple of 16"):
        VelocityTrackingEasyEnv(sim_device="cuda:0", headless=True, cfg=cfg, eval_cfg=ev)


def test_eval_environments_on_their_own_terrain_region(monkeypatch):
    """generated terrain with `eval_cfg` (reference legged_robot.py:502-503, terrain.py:37-54): the evaluation tile grid lies
    behind the training one in the same height field, the evaluation environments start on its tiles, and their teleport
    window is shifted by the region's offset (`_teleport_robots` :1033)."""
    import fake_sim
    from go1_gym.envs.go1.velocity_tracking import VelocityTrackingEasyEnv
    fake_sim.install(monkeypatch)
    NT, NE = 32, 16
    cfg, ev = _cfgs(NT, NE)
    for c, rows, cols in ((cfg, 3, 4), (ev, 2, 6)):
        t = c.terrain
        t.mesh_type, t.num_rows, t.num_cols, t.terrain_length, t.terrain_width, t.border_size = "heightfield", rows, cols, 4.0, 4.0, 2.0
        t.terrain_proportions, t.curriculum, t.teleport_robots, t.teleport_thresh = [0.2, 0.2, 0.2, 0.2, 0.2], True, True, 0.3
        t.min_init_terrain_level, t.max_init_terrain_level, t.center_robots = 0, rows - 1, False
    torch.manual_seed(0)
    env = VelocityTrackingEasyEnv(sim_device="cuda:0", headless=True, cfg=cfg, eval_cfg=ev)
    tr_rows = 3 * 40 + 40
    assert env.terrain.heightsamples.shape == (tr_rows + 2 * 40 + 40, max(4 * 40 + 40, 6 * 40 + 40))
    assert (ev.terrain.x_offset, ev.terrain.rows_offset) == (tr_rows, 3)
    S, Se = env.sim_config, env.sim_config_eval
    assert (S.teleport_x_offset, Se.teleport_x_offset) == (0.0, float(int(tr_rows * 0.1)))
    assert (Se.terrain_num_rows, Se.terrain_num_cols, S.terrain_num_rows, S.terrain_num_cols) == (2, 6, 3, 4)
    assert S.custom_origins == 1 and Se.custom_origins == 1 and S.hf_rows == env.terrain.tot_rows
    ox = env.env_origins[:, 0]
    assert float(ox[:NT].max()) < 12.0 and float(ox[NT:].min()) > tr_rows * 0.1               # metres: behind the training region
    # every evaluation origin is one of eval_cfg's tile origins, at that tile's height
    tiles = torch.from_numpy(ev.terrain.env_origins).float().reshape(-1, 3)
    dist = (env.env_origins[NT:, None, :] - tiles[None]).abs().amax(-1).amin(-1)
    assert float(dist.max()) < 1e-5
    assert bool((env.terrain_types[NT:] == torch.div(torch.arange(NE), NE / 6, rounding_mode="floor").long()).all())
    env.step(torch.zeros(NT + NE, 12))
    assert bool(torch.isfinite(env.obs_buf).all())
    # the reset positions of the evaluation environments stay on their region
    assert float(env.root_states[NT:, 0].min()) > tr_rows * 0.1 - 1.0


def test_eval_terrain_needs_a_training_terrain(monkeypatch):
    import fake_sim
    from go1_gym.envs.go1.velocity_tracking import VelocityTrackingEasyEnv
    fake_sim.install(monkeypatch)
    cfg, ev = _cfgs(32, 16)
    ev.terrain.mesh_type = "heightfield"
    with pytest.raises(ValueError, match="appended to the training terrain"):
        VelocityTrackingEasyEnv(sim_device="cuda:0", headless=True, cfg=cfg, eval_cfg=ev)
