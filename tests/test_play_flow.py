"""scripts/play.py's flow through the product's host classes, fed the `parameters.pkl` of the reference's pretrained run
(tests/golden/pretrain_parameters.json, make_golden.py gen_pretrain_parameters): the configuration is applied the way
`load_env` applies it (play.py:37-47), the values the reference DERIVED at construction and stored in that file
(`_parse_cfg` legged_robot.py:1716-1754; `Terrain._load_cfg` / env origins, terrain.py:56-66,161-179) must come out the same
here, and the play loop (one environment, commands written into `env.commands` every step, play.py:62-139) runs.  No GPU: the
simulator handle is the oracle-backed stand-in of tests/fake_sim.py."""
import json
import os

import numpy as np
import pytest
import torch

from util import GOLDEN


def _apply_like_load_env(Cfg, stored):
    for key, value in stored.items():                 # play.py:43-46
        if hasattr(Cfg, key):
            for key2, value2 in value.items():
                setattr(getattr(Cfg, key), key2, value2)


def _pretrained():
    with open(os.path.join(GOLDEN, "pretrain_parameters.json")) as f:
        return json.load(f)


def test_train_config_equals_the_pretrained_runs_parameters():
    """scripts/train_config.py (the mirror of train.py:21-204) against what the reference's own run logged: every stored field
    is equal except the three the current train.py sets differently from that run (train.py:49,112-113) and the derived ones."""
    from go1_gym.envs.base.legged_robot_config import make_cfg
    from scripts.train_config import apply_train_config
    stored = _pretrained()["Cfg"]
    cfg = apply_train_config(make_cfg())
    derived = {"domain_rand": {"gravity_rand_duration", "gravity_rand_interval", "push_interval", "rand_interval"},
               "env": {"max_episode_length"},
               "terrain": {"border", "env_length", "env_origins", "env_width", "length_per_env_pixels", "max_terrain_level",
                           "num_sub_terrains", "proportions", "rows_offset", "terrain_origins", "tot_cols", "tot_rows",
                           "width_per_env_pixels", "x_offset"}}
    differs = {}
    for sec, vals in stored.items():
        if sec == "command_ranges":                   # = vars(cfg.commands), created by _parse_cfg (:1721)
            continue
        for k, v in vals.items():
            if k in derived.get(sec, ()) or (sec, k) == ("reward_scales", "jump_amplitude"):       # (a term the run's code had)
                continue
            m = getattr(getattr(cfg, sec), k)
            m = vars(m) if hasattr(m, "__dict__") and not isinstance(m, dict) else m
            m = list(m) if isinstance(m, tuple) else m
            if m != v:
                differs[(sec, k)] = (m, v)
    assert differs == {("domain_rand", "gravity_range"): ([-1.0, 1.0], [-2.0, 2.0]),
                       ("rewards", "terminal_body_ori"): (1.6, 0.5),
                       ("rewards", "use_terminal_roll_pitch"): (True, False)}, differs
    P = _pretrained()
    from go1_gym_learn.ppo_cse.actor_critic import AC_Args
    from go1_gym_learn.ppo_cse.ppo import PPO_Args
    from go1_gym_learn.ppo_cse import RunnerArgs
    for cls, name in ((AC_Args, "AC_Args"), (PPO_Args, "PPO_Args"), (RunnerArgs, "RunnerArgs")):
        for k, v in P[name].items():
            assert getattr(cls, k) == v, (name, k, getattr(cls, k), v)


def test_derived_configuration_and_play_loop(monkeypatch):
    import fake_sim
    from go1_gym.envs.base.legged_robot_config import make_cfg
    from go1_gym.envs.go1.go1_config import config_go1
    from go1_gym.envs.go1.velocity_tracking import VelocityTrackingEasyEnv
    from go1_gym.envs.wrappers.history_wrapper import HistoryWrapper
    fake_sim.install(monkeypatch)
    stored = _pretrained()["Cfg"]
    want = json.loads(json.dumps(stored))             # (the constructor overwrites the derived entries in place)

    # ---- the run's own configuration (fewer environments): the derived values must be the stored ones
    Cfg = make_cfg()
    config_go1(Cfg)
    _apply_like_load_env(Cfg, stored)
    Cfg.env.num_envs = 32
    torch.manual_seed(0)
    env = VelocityTrackingEasyEnv(sim_device="cuda:0", headless=True, cfg=Cfg)
    assert float(env.max_episode_length) == want["env"]["max_episode_length"] == 1001.0
    for k in ("push_interval", "rand_interval", "gravity_rand_interval", "gravity_rand_duration"):
        assert float(getattr(Cfg.domain_rand, k)) == want["domain_rand"][k], k
    for k in ("border", "env_length", "env_width", "length_per_env_pixels", "width_per_env_pixels", "num_sub_terrains", "tot_cols",
              "tot_rows", "x_offset", "rows_offset", "max_terrain_level"):
        assert getattr(Cfg.terrain, k) == want["terrain"][k], k
    assert [float(p) for p in Cfg.terrain.proportions] == [float(p) for p in want["terrain"]["proportions"]]
    np.testing.assert_array_equal(np.asarray(Cfg.terrain.env_origins), np.asarray(want["terrain"]["env_origins"]))
    np.testing.assert_array_equal(Cfg.terrain.terrain_origins.cpu().numpy(), np.asarray(want["terrain"]["terrain_origins"], dtype=np.float32))
    assert Cfg.command_ranges == vars(Cfg.commands) and Cfg.command_ranges["num_bins_vel_x"] == want["command_ranges"]["num_bins_vel_x"]
    # center_robots: the environments start on the central (2 * center_span)^2 tiles (:1686-1694)
    lv, ty = env.terrain_levels, env.terrain_types
    assert int(lv.min()) >= 11 and int(lv.max()) <= 18 and int(ty.min()) >= 11 and int(ty.max()) <= 18
    S = env.sim_config
    assert (S.gravity_rand_interval, S.gravity_rand_duration, S.rand_interval, S.max_episode_length) == (401, 397, 201, 1001)
    assert tuple(S.gravity_range) == (-2.0, 2.0) and S.lag_timesteps == 6 and S.hf_rows == 1500

    # ---- play.py:48-78 on top, then the loop of :120-139 with its command writes
    Cfg = make_cfg()
    config_go1(Cfg)
    _apply_like_load_env(Cfg, want)
    dr = Cfg.domain_rand
    dr.push_robots = dr.randomize_friction = dr.randomize_gravity = dr.randomize_restitution = dr.randomize_motor_offset = False
    dr.randomize_motor_strength = dr.randomize_friction_indep = dr.randomize_ground_friction = dr.randomize_base_mass = False
    dr.randomize_Kd_factor = dr.randomize_Kp_factor = dr.randomize_joint_friction = dr.randomize_com_displacement = False
    Cfg.env.num_recording_envs = 1
    Cfg.env.num_envs = 1
    Cfg.terrain.num_rows = Cfg.terrain.num_cols = 5
    Cfg.terrain.border_size = 0
    Cfg.terrain.center_robots = True
    Cfg.terrain.center_span = 1
    Cfg.terrain.teleport_robots = True
    dr.lag_timesteps = 6
    dr.randomize_lag_timesteps = True
    Cfg.control.control_type = "actuator_net"
    env = HistoryWrapper(VelocityTrackingEasyEnv(sim_device="cuda:0", headless=False, cfg=Cfg))
    assert env.num_envs == 1 and env.env.sim_config.teleport_robots == 1 and env.env.terrain.tot_rows == 250
    obs = env.reset()
    assert obs["obs"].shape == (1, 70) and obs["obs_history"].shape == (1, 2100) and obs["privileged_obs"].shape == (1, 2)
    gait = torch.tensor([0.5, 0.0, 0.0])
    for i in range(20):
        actions = torch.zeros(1, 12)
        env.commands[:, 0] = 1.5
        env.commands[:, 1] = 0.0
        env.commands[:, 2] = 0.0
        env.commands[:, 3] = 0.0
        env.commands[:, 4] = 3.0
        env.commands[:, 5:8] = gait
        env.commands[:, 8] = 0.5
        env.commands[:, 9] = 0.08
        env.commands[:, 10] = 0.0
        env.commands[:, 11] = 0.0
        env.commands[:, 12] = 0.25
        obs, rew, done, info = env.step(actions)
        assert float(env.base_lin_vel[0, 0]) == float(env.base_lin_vel[0, 0]) and env.dof_pos[0, :].cpu().shape == (12,)
    # the written commands are what the observation carries (scaled, legged_robot.py:325): they were not resampled away
    assert float(obs["obs"][0, 3]) == pytest.approx(1.5 * Cfg.obs_scales.lin_vel) and bool(torch.isfinite(obs["obs_history"]).all())
    assert float(obs["obs"][0, 3 + 4]) == pytest.approx(3.0 * Cfg.obs_scales.gait_freq_cmd)


def test_gravity_attributes_follow_the_simulators_schedule(monkeypatch):
    """`env.gravities` / `env.gravity_vec` (reference legged_robot.py:549-559) are evaluated on the host from (seed, step counter);
    the projected gravity the simulator returns for step k was rotated from the vector in force DURING that step (:104 precedes
    the callback that may change it, :701-705)."""
    import fake_sim
    import go1sim_host as H
    from go1_gym.envs.base.legged_robot_config import make_cfg
    from go1_gym.envs.go1.velocity_tracking import VelocityTrackingEasyEnv
    from go1_gym.utils.math_utils import quat_rotate_inverse
    from scripts.train_config import apply_train_config
    fake_sim.install(monkeypatch)
    cfg = apply_train_config(make_cfg(), num_envs=16)
    cfg.terrain.mesh_type = "plane"
    cfg.domain_rand.gravity_rand_interval_s, cfg.domain_rand.gravity_impulse_duration = 0.06, 0.67      # 3 steps: 2 on, 1 off
    env = VelocityTrackingEasyEnv(sim_device="cuda:0", headless=True, cfg=cfg)
    S = env.sim_config
    assert S.gravity_rand_duration < S.gravity_rand_interval <= 4
    env.reset()                                                    # (reset_idx of everything + one zero-action step, :241-246)
    k0 = env.common_step_counter
    seen_on = seen_off = 0
    for k in range(k0 + 1, k0 + 10):
        env.step(torch.zeros(16, 12))
        assert env.common_step_counter == k and env.gravities.shape == (16, 3) and env.gravity_vec.shape == (16, 3)
        during = torch.from_numpy(H.gravity_at(S, k - 1))
        want = quat_rotate_inverse(env.base_quat, (during / during.norm()).repeat(16, 1))
        keep = ~env.reset_buf.bool()                               # (a reset environment reports its re-initialised state)
        assert int(keep.sum()) >= 12 and float((env.projected_gravity - want)[keep].abs().max()) < 2e-5
        off = float(env.gravities.abs().max())
        seen_on += off > 0
        seen_off += off == 0
        assert float((env.gravity_vec.norm(dim=1) - 1).abs().max()) < 1e-6
    assert seen_on >= 3 and seen_off >= 1 and env.default_body_mass == pytest.approx(4.801)


def test_state_writes_are_the_push(monkeypatch):
    """Ownership / aliasing of SURVEY 8b: `root_states`, `dof_pos`, `commands` are the simulator's own state — a write
    (`set_idx_pose`, `set_main_agent_pose`, direct indexing; reference legged_robot.py:241-261, 525-528) is what the next step
    starts from; `reset_idx` of nothing is a no-op and of a subset touches only that subset (:150-153)."""
    import fake_sim
    from go1_gym.envs.base.legged_robot_config import make_cfg
    from go1_gym.envs.go1.velocity_tracking import VelocityTrackingEasyEnv
    from scripts.train_config import apply_train_config
    fake_sim.install(monkeypatch)
    cfg = apply_train_config(make_cfg(), num_envs=16)
    cfg.terrain.mesh_type = "plane"
    cfg.domain_rand.randomize_gravity = False
    cfg.commands.resampling_time = 1000.0
    env = VelocityTrackingEasyEnv(sim_device="cuda:0", headless=True, cfg=cfg)
    env.reset()
    for _ in range(3):
        env.step(torch.zeros(16, 12))
    ids = torch.tensor([3, 9, 0])
    pose = env.default_dof_pos.repeat(3, 1)
    base = torch.zeros(3, 13)
    base[:, 2], base[:, 6] = 1.0, 1.0                 # 1 m up, identity quaternion, at rest
    base[:, 0:2] = env.root_states[ids, 0:2]
    env.set_idx_pose(ids, pose, base)
    env.set_main_agent_pose([0.5, -0.5, 2.0], [0.0, 0.0, 0.0, 1.0])
    env.root_states[0, 7:13] = 0.0
    env.commands[:, 0] = 0.7
    before_len = env.episode_length_buf.clone()
    env.step(torch.zeros(16, 12))
    h = 0.02
    for i, z0 in ((3, 1.0), (9, 1.0), (0, 2.0)):      # free fall for one policy step (4 substeps, semi-implicit Euler)
        assert float(env.root_states[i, 2]) == pytest.approx(z0 - 0.5 * 9.8 * h * h * (1 + 1 / 4), abs=2e-3), i
        assert float(env.root_states[i, 9]) == pytest.approx(-9.8 * h, abs=0.02)      # (base origin, not COM: the legs are moving)
    assert float(env.root_states[0, 0]) == pytest.approx(0.5, abs=1e-3) and float(env.root_states[0, 1]) == pytest.approx(-0.5, abs=1e-3)
    assert float((env.dof_pos[ids] - pose).abs().max()) < 0.05          # (held at the written pose by the actuators)
    assert float(env.obs_buf[5, 3]) == pytest.approx(0.7 * cfg.obs_scales.lin_vel)
    assert bool((env.episode_length_buf == before_len + 1).all())
    env.reset_idx(torch.tensor([], dtype=torch.long))
    assert bool((env.episode_length_buf == before_len + 1).all())
    env.reset_idx(torch.tensor([4, 11]))
    assert [int(v) for v in env.episode_length_buf[[4, 11]]] == [0, 0] and int((env.episode_length_buf == 0).sum()) == 2


def test_history_wrapper_slots_match_reference(monkeypatch):
    """which observation sits in which history slot after every call, against the reference `HistoryWrapper` driven through the
    same call script over a mock environment (history_trace.json): oldest first, the extra shift of `get_observations`
    (history_wrapper.py:29), everything cleared by `reset` (:40).  Here the history is a window of the ring the simulator
    appends to, not a concatenation."""
    import fake_sim
    from go1_gym.envs.base.legged_robot_config import make_cfg
    from go1_gym.envs.go1.velocity_tracking import VelocityTrackingEasyEnv
    from go1_gym.envs.wrappers.history_wrapper import HistoryWrapper
    from scripts.train_config import apply_train_config
    with open(os.path.join(GOLDEN, "history_trace.json")) as f:
        ref = json.load(f)
    fake_sim.install(monkeypatch)
    cfg = apply_train_config(make_cfg(), num_envs=16)
    cfg.terrain.mesh_type = "plane"
    cfg.env.num_observation_history = H = 4
    env = HistoryWrapper(VelocityTrackingEasyEnv(sim_device="cuda:0", headless=True, cfg=cfg))
    no = env.num_obs
    seen = [torch.zeros(16, no)]                       # observation k of the run (0 = an empty slot)
    g = torch.Generator().manual_seed(0)
    for call, want in zip(ref["script"], ref["trace"]):
        if call == "reset":
            out = env.reset()
            seen.append(out["obs"].clone())
        elif call == "step":
            out = env.step(torch.randn(16, 12, generator=g))[0]
            seen.append(out["obs"].clone())
        else:
            out = env.get_observations()
            assert torch.equal(out["obs"], seen[-1])
        hist = out["obs_history"].reshape(16, H, no)
        for slot, k in enumerate(want[0]):
            assert torch.equal(hist[:, slot], seen[k]), (call, slot, k)


def test_step_extras_carry_the_reference_keys_and_values(monkeypatch):
    """`VelocityTrackingEasyEnv.step` (velocity_tracking/__init__.py:22-44): the thirteen entries it adds to `extras`, each equal
    to the expression the reference builds it from (numpy on the host, read lazily here), and the 4-tuple it returns."""
    import fake_sim
    from go1_gym.envs.base.legged_robot_config import make_cfg
    from go1_gym.envs.go1.velocity_tracking import VelocityTrackingEasyEnv
    from scripts.train_config import apply_train_config
    fake_sim.install(monkeypatch)
    cfg = apply_train_config(make_cfg(), num_envs=16)
    cfg.terrain.mesh_type = "plane"
    env = VelocityTrackingEasyEnv(sim_device="cuda:0", headless=True, cfg=cfg)
    env.reset()
    ret = env.step(0.3 * torch.randn(16, 12, generator=torch.Generator().manual_seed(1)))
    assert len(ret) == 4 and ret[0] is env.obs_buf and ret[1] is env.rew_buf and ret[2] is env.reset_buf
    ex = ret[3]
    want = {"joint_pos": env.dof_pos, "joint_vel": env.dof_vel, "joint_pos_target": env.joint_pos_target,
            "body_linear_vel": env.base_lin_vel, "body_angular_vel": env.base_ang_vel, "body_linear_vel_cmd": env.commands[:, 0:2],
            "body_angular_vel_cmd": env.commands[:, 2:], "contact_states": env.contact_forces[:, env.feet_indices, 2] > 1.0,
            "foot_positions": env.foot_positions, "body_pos": env.root_states[:, 0:3], "torques": env.torques}
    for k, t in want.items():
        v = ex[k]
        assert isinstance(v, np.ndarray) and v.shape == tuple(t.shape), k
        assert np.array_equal(v, t.cpu().numpy()), k
    assert torch.equal(ex["joint_vel_target"], torch.zeros(12)) and ex["privileged_obs"] is env.privileged_obs_buf
    assert ex["contact_states"].dtype == np.bool_ and ex["foot_positions"].shape == (16, 4, 3)
    assert {"env_bins", "time_outs", "train/episode"} <= set(ex)


def test_initial_dynamics_dict_presets(monkeypatch):
    """`initial_dynamics_dict` (velocity_tracking/__init__.py:11, legged_robot.py:1283-1288): preset per-environment dynamics
    parameters survive construction for the quantities whose randomisation switch is off and are re-drawn for those whose
    switch is on (the set-up draw of :1548 comes after the preset); motor strengths last until the first reset re-draws them."""
    import fake_sim
    from go1_gym.envs.base.legged_robot_config import make_cfg
    from go1_gym.envs.go1.velocity_tracking import VelocityTrackingEasyEnv
    from scripts.train_config import apply_train_config
    fake_sim.install(monkeypatch)
    N = 16
    cfg = apply_train_config(make_cfg(), num_envs=N)
    cfg.terrain.mesh_type = "plane"
    dr = cfg.domain_rand
    dr.randomize_friction, dr.randomize_restitution, dr.randomize_base_mass = False, True, False
    preset = dict(friction_coeffs=torch.full((N, 4), 0.77), restitutions=torch.full((N, 4), 0.9), payloads=torch.full((N,), 2.5),
                  com_displacements=torch.full((N, 3), 0.05), motor_strengths=torch.full((N, 12), 1.07),
                  Kp_factors=torch.full((N, 12), 1.2), Kd_factors=torch.full((N, 12), 0.8))
    torch.manual_seed(0)
    env = VelocityTrackingEasyEnv(sim_device="cuda:0", headless=True, cfg=cfg, initial_dynamics_dict=preset)
    B = env.buffers
    assert bool((B.friction_coeffs == 0.77).all()) and bool((B.payloads == 2.5).all())                 # switches off: kept
    assert bool((B.restitutions <= 0.4).all()) and not bool((B.restitutions == 0.9).any())              # switch on: re-drawn in range
    assert bool((env.com_displacements == 0.05).all()) and bool((env.Kp_factors == 1.2).all()) and bool((env.Kd_factors == 0.8).all())
    assert bool((env.motor_strengths == 1.07).all())
    assert env.friction_coeffs.shape == (N, 4) and float(env.friction_coeffs[3, 2]) == pytest.approx(0.77)
    env.reset()                                           # reset_idx re-draws the DOF properties (:164)
    assert not bool((env.motor_strengths == 1.07).any()) and bool(((env.motor_strengths >= 0.9) & (env.motor_strengths <= 1.1)).all())
    assert bool((B.friction_coeffs == 0.77).all())
