#!/usr/bin/env python3
"""Generate golden input/output vectors by executing the REFERENCE's own Python on CPU.

Runs only in the authoring container (needs /root/reference).  Technique: inject
test-only stub modules for the packages the reference imports but that are absent here
(`isaacgym`, `params_proto`, `ml_logger`, `gym`), import go1_gym.envs.base.legged_robot and
go1_gym.envs.rewards.corl_rewards FROM /root/reference, and call the unbound methods
(`LeggedRobot._compute_torques`, `_step_contact_targets`, `check_termination`, `compute_reward`,
`compute_observations`, `_get_noise_scale_vec`, `_prepare_reward_function`) on a mock object that
carries synthetic state tensors.  No reference code is copied; its functions are executed.
The `isaacgym.torch_utils` helpers the reference calls are plain quaternion algebra and come from
walk-these-ways_amd/go1_gym/utils/math_utils.py (loaded under a private name).

Output: tests/golden/maps_<variant>.npz, torques_<variant>.npz, curriculum.npz
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
PKG = os.path.join(REPO, "walk-these-ways_amd")


def load_private(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def install_stubs():
    sys.path.insert(0, os.path.join(PKG, "shims"))          # params_proto stand-in
    mu = load_private("_wtw_math_utils", os.path.join(PKG, "go1_gym", "utils", "math_utils.py"))
    isaacgym = types.ModuleType("isaacgym")
    tu = types.ModuleType("isaacgym.torch_utils")
    for n in ("quat_rotate_inverse", "quat_apply", "quat_from_angle_axis", "quat_mul", "quat_conjugate", "normalize",
              "torch_rand_float", "to_torch", "get_axis_params", "quat_rotate"):
        setattr(tu, n, getattr(mu, n))
    tu.torch = torch
    tu.np = np
    for n in ("gymtorch", "gymapi", "gymutil", "terrain_utils"):
        m = types.ModuleType(f"isaacgym.{n}")
        setattr(isaacgym, n, m)
        sys.modules[f"isaacgym.{n}"] = m
    isaacgym.torch_utils = tu
    sys.modules["isaacgym"] = isaacgym
    sys.modules["isaacgym.torch_utils"] = tu
    gym = types.ModuleType("gym")
    gym.Env = object
    gym.Wrapper = object
    gym.spaces = types.ModuleType("gym.spaces")
    sys.modules["gym"] = gym
    sys.modules["gym.spaces"] = gym.spaces
    sys.path.insert(0, REF)


class Mock:
    pass


def rand_quat(rng, n, tilt=0.5):
    ax = rng.standard_normal((n, 3))
    ax /= np.linalg.norm(ax, axis=1, keepdims=True)
    ang = rng.uniform(-tilt, tilt, n)
    yaw = rng.uniform(-3.14, 3.14, n)
    q1 = np.concatenate([ax * np.sin(ang / 2)[:, None], np.cos(ang / 2)[:, None]], 1)
    q2 = np.stack([0 * yaw, 0 * yaw, np.sin(yaw / 2), np.cos(yaw / 2)], 1)
    from _wtw_math_utils import quat_mul
    return quat_mul(torch.tensor(q2), torch.tensor(q1)).float()


def make_env(variant, N, seed, mild=False):
    """A mock LeggedRobot carrying a random but plausible post-physics state."""
    from go1_gym.envs.base.legged_robot import LeggedRobot       # the REFERENCE module
    from go1_gym.envs.base.legged_robot_config import Cfg
    from _wtw_math_utils import quat_rotate_inverse
    tc = load_private("_wtw_train_config", os.path.join(PKG, "scripts", "train_config.py"))
    var = load_private("_wtw_variants", os.path.join(HERE, "variants.py"))
    tc.apply_train_config(Cfg)
    var.apply_variant(Cfg, variant)
    rng = np.random.default_rng(seed)
    f = lambda *s, lo=-1.0, hi=1.0: torch.tensor(rng.uniform(lo, hi, s), dtype=torch.float)
    k = 0.08 if mild else 1.0        # "mild" states sit near nominal standing so the total reward is not ~0

    e = Mock()
    e.cfg = Cfg
    e.device = "cpu"
    e.num_envs = e.num_train_envs = N
    e.num_actions = e.num_dof = e.num_dofs = e.num_actuated_dof = 12
    e.num_bodies = 17
    e.sim_params = Mock()
    e.sim_params.dt = float(np.float32(Cfg.sim.dt))
    LeggedRobot._parse_cfg(e, Cfg)
    legs = ["FL", "FR", "RL", "RR"]
    e.dof_names = [f"{l}_{p}_joint" for l in legs for p in ("hip", "thigh", "calf")]
    e.default_dof_pos = torch.tensor([Cfg.init_state.default_joint_angles[n] for n in e.dof_names]).unsqueeze(0)
    e.feet_indices = torch.tensor([4, 8, 12, 16])
    e.penalised_contact_indices = torch.tensor([2, 6, 10, 14, 3, 7, 11, 15])
    e.termination_contact_indices = torch.tensor([0])
    lo = torch.tensor([-0.802851455917, -1.0471975512, -2.69653369433] * 4)
    hi = torch.tensor([0.802851455917, 4.18879020479, -0.916297857297] * 4)
    m, r = (lo + hi) / 2, hi - lo
    e.dof_pos_limits = torch.stack([m - 0.5 * r * Cfg.rewards.soft_dof_pos_limit, m + 0.5 * r * Cfg.rewards.soft_dof_pos_limit], 1)
    e.torque_limits = torch.full((12,), 33.5)
    e.p_gains = torch.full((12,), 20.0)
    e.d_gains = torch.full((12,), 0.5)

    # --- state
    e.root_states = torch.zeros(N, 13)
    e.root_states[:, 0:2] = f(N, 2, lo=-3, hi=3)
    e.root_states[:, 2] = f(N, lo=0.25, hi=0.34) if mild else f(N, lo=0.03, hi=0.40)
    e.root_states[:, 3:7] = rand_quat(rng, N, tilt=0.5 * k)
    e.root_states[:, 7:13] = f(N, 6, lo=-1.5, hi=1.5) * (0.3 if mild else 1.0)
    e.base_pos = e.root_states[:, 0:3]
    e.base_quat = e.root_states[:, 3:7]
    e.dof_pos = e.default_dof_pos + f(N, 12, lo=-0.9, hi=0.9) * k
    e.dof_vel = f(N, 12, lo=-8, hi=8) * k
    e.gravity_vec = torch.tensor([0.3, -0.2, -9.8]).div(torch.tensor([0.3, -0.2, -9.8]).norm()).repeat(N, 1)
    e.gravities = torch.tensor([0.3, -0.2, 0.0]).repeat(N, 1)
    e.base_lin_vel = quat_rotate_inverse(e.base_quat, e.root_states[:, 7:10])
    e.base_ang_vel = quat_rotate_inverse(e.base_quat, e.root_states[:, 10:13])
    e.projected_gravity = quat_rotate_inverse(e.base_quat, e.gravity_vec)
    e.foot_positions = e.base_pos.unsqueeze(1) + f(N, 4, 3, lo=-0.4, hi=0.4)
    if mild:
        from _wtw_math_utils import quat_apply_yaw
        nominal = torch.tensor([[0.2, 0.14, 0.], [0.2, -0.14, 0.], [-0.2, 0.14, 0.], [-0.2, -0.14, 0.]]).repeat(N, 1, 1)
        nominal = nominal + f(N, 4, 3, lo=-0.03, hi=0.03)
        for i in range(4):
            e.foot_positions[:, i] = e.base_pos + quat_apply_yaw(e.base_quat.clone(), nominal[:, i])
    e.foot_positions[:, :, 2] = f(N, 4, lo=0.0, hi=0.15) * (0.5 if mild else 1.0) + (0.02 if mild else 0.0)
    e.foot_velocities = f(N, 4, 3, lo=-2, hi=2) * k
    e.prev_foot_velocities = f(N, 4, 3, lo=-2, hi=2)
    cf = f(N, 17, 3, lo=-1, hi=1) * torch.tensor(rng.choice([0.0, 0.05, 2.0, 60.0, 200.0], (N, 17, 1)), dtype=torch.float)
    cf[:, :, 2] = cf[:, :, 2].abs()
    if mild:
        cf[:, [0, 1, 2, 3, 5, 6, 7, 9, 10, 11, 13, 14, 15]] = 0.0
        cf[:, [4, 8, 12, 16]] = cf[:, [4, 8, 12, 16]].clamp(-40, 40)
    cf[: N // 2, 0] *= 0.0                                     # half the envs: no base contact -> no termination
    e.contact_forces = cf
    e.actions = f(N, 12, lo=-3, hi=3) * k
    e.last_actions = k * f(N, 12, lo=-3, hi=3) * torch.tensor(rng.choice([0.0, 1.0], (N, 12), p=[0.2, 0.8]), dtype=torch.float)
    e.last_last_actions = k * f(N, 12, lo=-3, hi=3) * torch.tensor(rng.choice([0.0, 1.0], (N, 12), p=[0.2, 0.8]), dtype=torch.float)
    e.joint_pos_target = e.default_dof_pos + f(N, 12, lo=-0.7, hi=0.7) * k
    e.last_joint_pos_target = e.default_dof_pos + f(N, 12, lo=-0.7, hi=0.7) * k
    e.last_last_joint_pos_target = e.default_dof_pos + f(N, 12, lo=-0.7, hi=0.7) * k
    e.last_dof_vel = e.dof_vel + f(N, 12, lo=-8, hi=8) * k * 0.1 if mild else f(N, 12, lo=-8, hi=8)
    e.torques = f(N, 12, lo=-33.5, hi=33.5) * (0.2 if mild else 1.0)
    e.last_contacts = torch.tensor(rng.choice([False, True], (N, 4)))
    c = Cfg.commands
    rngs = [c.lin_vel_x, c.lin_vel_y, c.ang_vel_yaw, c.body_height_cmd, c.gait_frequency_cmd_range, [0, 1], [0, 1], [0, 1],
            c.gait_duration_cmd_range, c.footswing_height_range, c.body_pitch_range, [-0.2, 0.2], c.stance_width_range,
            c.stance_length_range, c.aux_reward_coef_range]
    e.commands = torch.stack([f(N, lo=a, hi=b) if b > a else torch.full((N,), float(a)) for a, b in rngs], 1)
    e.commands[:, 5:8] = torch.round(2 * e.commands[:, 5:8]) / 2.0 % 1
    e.commands[N // 4: N // 2, 8] = f(N // 2 - N // 4, lo=0.3, hi=0.7)     # non-default stance durations too
    e.gait_indices = f(N, lo=0, hi=1)
    e.clock_inputs = torch.zeros(N, 4)
    e.doubletime_clock_inputs = torch.zeros(N, 4)
    e.halftime_clock_inputs = torch.zeros(N, 4)
    e.desired_contact_states = torch.zeros(N, 4)
    e.episode_length_buf = torch.tensor(rng.integers(2, 480, N), dtype=torch.long)
    e.episode_length_buf[-3:] = int(e.cfg.env.max_episode_length) + torch.tensor([0, 1, 2])
    e.measured_heights = 0
    # domain randomisation
    e.friction_coeffs = f(N, 1, lo=0.1, hi=3.0).repeat(1, 4)
    e.restitutions = f(N, 1, lo=0.0, hi=0.4).repeat(1, 4)
    e.payloads = f(N, lo=-1, hi=3)
    e.com_displacements = f(N, 3, lo=-0.1, hi=0.1)
    e.motor_strengths = f(N, 1, lo=0.9, hi=1.1).repeat(1, 12)
    e.motor_offsets = f(N, 12, lo=-0.02, hi=0.02)
    e.Kp_factors = f(N, 1, lo=0.8, hi=1.3).repeat(1, 12)
    e.Kd_factors = f(N, 1, lo=0.5, hi=1.5).repeat(1, 12)
    os_ = e.obs_scales
    e.commands_scale = torch.tensor([os_.lin_vel, os_.lin_vel, os_.ang_vel, os_.body_height_cmd, os_.gait_freq_cmd,
                                     os_.gait_phase_cmd, os_.gait_phase_cmd, os_.gait_phase_cmd, os_.gait_phase_cmd,
                                     os_.footswing_height_cmd, os_.body_pitch_cmd, os_.body_roll_cmd, os_.stance_width_cmd,
                                     os_.stance_length_cmd, os_.aux_reward_cmd])[:Cfg.commands.num_commands]
    e.forward_vec = torch.tensor([1., 0., 0.]).repeat(N, 1)
    e.obs_buf = torch.zeros(N, Cfg.env.num_observations)
    e.noise_scale_vec = LeggedRobot._get_noise_scale_vec(e, Cfg)
    e.rew_buf = torch.zeros(N)
    e.rew_buf_pos = torch.zeros(N)
    e.rew_buf_neg = torch.zeros(N)
    LeggedRobot._prepare_reward_function(e)
    for k in e.episode_sums:
        e.episode_sums[k] = f(N, lo=-2, hi=2)
    for k in e.command_sums:
        e.command_sums[k] = f(N, lo=-2, hi=2)
    return e, LeggedRobot


def flat(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


def gen_maps(variant, N=48, seed=11, mild=False, noise=None):
    """`noise` = (sim_seed, step): observation noise on — `torch.rand_like(obs_buf)` (legged_robot.py:375-376) returns the uniforms
    the oracle / kernel draw for (sim_seed, env, step, purpose 1, column)."""
    e, LR = make_env(variant, N, seed, mild)
    inp = dict(root_states=e.root_states.clone(), dof_pos=e.dof_pos.clone(), dof_vel=e.dof_vel.clone(),
               gravity=torch.tensor([0.3, -0.2, -9.8]), foot_positions=e.foot_positions.clone(),
               foot_velocities=e.foot_velocities.clone(), prev_foot_velocities=e.prev_foot_velocities.clone(),
               contact_forces=e.contact_forces.clone(), actions=e.actions.clone(), last_actions=e.last_actions.clone(),
               last_last_actions=e.last_last_actions.clone(), joint_pos_target=e.joint_pos_target.clone(),
               last_joint_pos_target=e.last_joint_pos_target.clone(),
               last_last_joint_pos_target=e.last_last_joint_pos_target.clone(), last_dof_vel=e.last_dof_vel.clone(),
               torques=e.torques.clone(), last_contacts=e.last_contacts.clone(), commands=e.commands.clone(),
               gait_indices=e.gait_indices.clone(), episode_length_buf=e.episode_length_buf.clone(),
               friction_coeffs=e.friction_coeffs[:, 0].clone(), restitutions=e.restitutions[:, 0].clone(),
               payloads=e.payloads.clone(), com_displacements=e.com_displacements.clone(),
               motor_strengths=e.motor_strengths.clone(), motor_offsets=e.motor_offsets.clone(),
               episode_sums=torch.stack([e.episode_sums[k] for k in e.episode_sums]).clone(),
               command_sums=torch.stack([e.command_sums[k] for k in e.command_sums]).clone())
    names = dict(episode_sum_names=np.array(list(e.episode_sums)), command_sum_names=np.array(list(e.command_sums)),
                 reward_names=np.array(e.reward_names),
                 reward_scales=np.array([e.reward_scales[n] for n in e.reward_names]))
    # the reference's post_physics_step order (legged_robot.py:117-124), physics-free parts
    LR._step_contact_targets(e)
    LR.check_termination(e)
    LR.compute_reward(e)
    extra = {}
    if noise is None:
        LR.compute_observations(e)
    else:
        sim_seed, step = noise
        assert e.cfg.noise.add_noise
        e.add_noise = True
        real = torch.rand_like
        torch.rand_like = lambda t, **k: torch.tensor([[philox_uniform(sim_seed, i, step, 1, c) for c in range(t.shape[1])]
                                                       for i in range(t.shape[0])])
        try:
            LR.compute_observations(e)
        finally:
            torch.rand_like = real
        extra = dict(sim_seed=np.array(sim_seed), step=np.array(step))
    clip = e.cfg.normalization.clip_observations
    out = dict(out_gait_indices=e.gait_indices, out_foot_indices=e.foot_indices, out_clock_inputs=e.clock_inputs,
               out_desired_contact_states=e.desired_contact_states, out_reset_buf=e.reset_buf, out_time_out_buf=e.time_out_buf,
               out_rew_buf=e.rew_buf, out_last_contacts=e.last_contacts,
               out_episode_sums=torch.stack([e.episode_sums[k] for k in e.episode_sums]),
               out_command_sums=torch.stack([e.command_sums[k] for k in e.command_sums]),
               out_obs=torch.clip(e.obs_buf, -clip, clip), out_priv=torch.clip(e.privileged_obs_buf, -clip, clip),
               out_noise_scale_vec=e.noise_scale_vec, out_base_lin_vel=e.base_lin_vel, out_base_ang_vel=e.base_ang_vel,
               out_projected_gravity=e.projected_gravity,
               out_max_episode_length=np.array(int(e.cfg.env.max_episode_length)),
               out_dof_pos_soft_limits=e.dof_pos_limits)
    np.savez_compressed(os.path.join(HERE, f"maps_{variant}{'_mild' if mild else ''}.npz"), **flat(inp), **flat(out), **names, **extra)
    print("maps", variant, "rew mean", float(e.rew_buf.mean()), "resets", int(e.reset_buf.sum()))


def gen_maps_fuzz():
    """tensor maps under the random switch sets of variants.py (FUZZ_VARIANTS)"""
    var = load_private("_wtw_variants_count", os.path.join(HERE, "variants.py"))
    for k in range(var.FUZZ_VARIANTS):
        for m in [x for x in sys.modules if x.startswith("go1_gym")]:
            del sys.modules[m]
        gen_maps(f"fuzz{k}", seed=60 + k, mild=True)


def gen_torques(variant, N=16, steps=12, seed=5):
    e, LR = make_env(variant, N, seed)
    net = torch.jit.load(os.path.join(REF, "resources/actuator_nets/unitree_go1.pt"), map_location="cpu")

    def eval_net(p, pl, pll, v, vl, vll):     # same packing as legged_robot.py:1242-1251 (closure not importable)
        xs = torch.stack((p, pl, pll, v, vl, vll), dim=-1)
        return net(xs.view(N * 12, 6)).view(N, 12)
    e.actuator_network = eval_net
    rng = np.random.default_rng(seed + 1)
    e.lag_buffer = [torch.zeros(N, 12) for _ in range(e.cfg.domain_rand.lag_timesteps + 1)]
    for n in ("joint_pos_err_last_last", "joint_pos_err_last", "joint_vel_last_last", "joint_vel_last"):
        setattr(e, n, torch.zeros(N, 12))
    rec = dict(motor_strengths=e.motor_strengths, motor_offsets=e.motor_offsets, Kp_factors=e.Kp_factors, Kd_factors=e.Kd_factors)
    acts, qs, qds, taus, tgts = [], [], [], [], []
    for s in range(steps):
        a = torch.tensor(rng.uniform(-4, 4, (N, 12)), dtype=torch.float)
        e.dof_pos = e.default_dof_pos + torch.tensor(rng.uniform(-0.8, 0.8, (N, 12)), dtype=torch.float)
        e.dof_vel = torch.tensor(rng.uniform(-10, 10, (N, 12)), dtype=torch.float)
        with torch.no_grad():
            tau = LR._compute_torques(e, a)
        acts.append(a); qs.append(e.dof_pos.clone()); qds.append(e.dof_vel.clone()); taus.append(tau.clone()); tgts.append(e.joint_pos_target.clone())
    rec.update(actions=torch.stack(acts), dof_pos=torch.stack(qs), dof_vel=torch.stack(qds), torques=torch.stack(taus),
               joint_pos_target=torch.stack(tgts))
    np.savez_compressed(os.path.join(HERE, f"torques_{variant}.npz"), **flat(rec))
    print("torques", variant, "mean |tau|", float(torch.stack(taus).abs().mean()))


def gen_curriculum():
    """reference curriculum.py (importable standalone): grid, set_to, get_local_bins, update."""
    from go1_gym.envs.base.curriculum import RewardThresholdCurriculum as Ref
    kw = dict(x_vel=(-5.0, 5.0, 21), y_vel=(-0.6, 0.6, 1), yaw_vel=(-5.0, 5.0, 21), body_height=(-0.25, 0.15, 2),
              gait_frequency=(2.0, 4.0, 3))
    r = Ref(seed=100, **kw)
    low, high = np.array([-1.0, -0.6, -1.0, -0.25, 2.0]), np.array([1.0, 0.6, 1.0, 0.15, 4.0])
    r.set_to(low=low, high=high)
    w0 = r.weights.copy()
    bins = np.array([5, 300, 301, 300, 900, 17])
    rew = [torch.tensor([1.0, 0.9, 0.2, 0.95, 0.99, 0.1]), torch.tensor([0.8, 0.9, 0.9, 0.1, 0.9, 0.9])]
    lr = np.array([0.55, 0.55, 0.55, 0.55, 0.35])
    r.update(bins, rew, [0.5, 0.5], local_range=lr)
    local = r.get_local_bins(np.array([0, 300, 1322]), ranges=lr)
    samples, inds = r.sample(64)
    np.savez_compressed(os.path.join(HERE, "curriculum.npz"), grid=r.grid, weights0=w0, weights1=r.weights, bins=bins,
                        rew0=rew[0].numpy(), rew1=rew[1].numpy(), local=local, samples=samples, inds=inds, low=low, high=high,
                        local_range=lr)
    print("curriculum weights", w0.sum(), r.weights.sum())


def gen_heights(seed=9, N=24, name="heights.npz", border=2.0, hscale=0.1, vscale=0.005, rows=120, cols=90, points=None):
    """reference LeggedRobot._init_height_points / _get_heights on a random height field (`points`: other measured_points_x / _y)."""
    from go1_gym.envs.base.legged_robot import LeggedRobot
    from go1_gym.envs.base.legged_robot_config import Cfg
    rng = np.random.default_rng(seed)
    e = Mock()
    e.device = "cpu"
    e.num_envs = N
    Cfg.terrain.mesh_type = "heightfield"
    Cfg.terrain.border_size = border
    Cfg.terrain.horizontal_scale = hscale
    Cfg.terrain.vertical_scale = vscale
    if points is not None:
        Cfg.terrain.measured_points_x, Cfg.terrain.measured_points_y = points
    e.terrain = Mock()
    e.terrain.cfg = Cfg.terrain
    e.height_samples = torch.tensor(rng.integers(-60, 60, (rows, cols)), dtype=torch.int16)
    e.root_states = torch.zeros(N, 13)
    e.root_states[:, 0] = torch.tensor(rng.uniform(-3.0, rows * hscale - border + 1.0, N), dtype=torch.float)     # some scans leave the map
    e.root_states[:, 1] = torch.tensor(rng.uniform(-3.0, cols * hscale - border + 1.0, N), dtype=torch.float)
    e.root_states[:, 2] = 0.4
    e.root_states[:, 3:7] = rand_quat(rng, N, tilt=0.3)
    e.base_quat = e.root_states[:, 3:7]
    ids = torch.arange(N)
    e.height_points = LeggedRobot._init_height_points(e, ids, Cfg)
    h = LeggedRobot._get_heights(e, ids, Cfg)
    np.savez_compressed(os.path.join(HERE, name), height_samples=e.height_samples.numpy(), root_states=e.root_states.numpy(),
                        heights=h.numpy(), points_x=np.array(Cfg.terrain.measured_points_x), points_y=np.array(Cfg.terrain.measured_points_y),
                        border=np.array(border), hscale=np.array(hscale), vscale=np.array(vscale))
    print("heights", h.shape, float(h.min()), float(h.max()))


def gen_ppo_fuzz():
    """the same fixed-rollout update under four other settings of PPO_Args (ppo.py:11-31): fixed / adaptive schedule, plain value
    loss, several adaptation sub-steps, selective adaptation loss, other clip / entropy / value coefficients, epochs, batches"""
    PPO_FUZZ = load_private("_wtw_variants_ppo", os.path.join(HERE, "variants.py")).PPO_FUZZ      # shared with tests/test_gpu_ppo_fused.py
    for k, over in enumerate(PPO_FUZZ):
        for m in [x for x in sys.modules if x.startswith("go1_gym")]:
            del sys.modules[m]
        gen_ppo(seed=10 + k, over=over, name=f"ppo_fuzz{k}.npz")


def gen_ppo(seed=3, over=None, name="ppo.npz"):
    """reference go1_gym_learn.ppo_cse: RolloutStorage.compute_returns + PPO.update on a fixed rollout."""
    ml = types.ModuleType("ml_logger")
    ml.logger = object()
    sys.modules["ml_logger"] = ml
    from go1_gym_learn.ppo_cse.actor_critic import ActorCritic, AC_Args
    from go1_gym_learn.ppo_cse.ppo import PPO, PPO_Args
    AC_Args.actor_hidden_dims = [32, 16]
    AC_Args.critic_hidden_dims = [24, 16]
    AC_Args.adaptation_module_branch_hidden_dims = [16, 8]
    for k_, v_ in (over or {}).items():
        assert hasattr(PPO_Args, k_), k_
        setattr(PPO_Args, k_, v_)
    N, T, no, npv, H, na = 20, 6, 10, 2, 3, 12
    torch.manual_seed(seed)
    ac = ActorCritic(no, npv, no * H, na)
    init = {k: v.clone() for k, v in ac.state_dict().items()}
    alg = PPO(ac, device="cpu")
    alg.init_storage(N, T, [no], [npv], [no * H], [na])
    st = alg.storage
    g = torch.Generator().manual_seed(seed + 1)
    r = lambda *s: torch.randn(*s, generator=g)
    st.observations.copy_(r(T, N, no)); st.privileged_observations.copy_(r(T, N, npv))
    st.observation_histories.copy_(r(T, N, no * H)); st.actions.copy_(r(T, N, na))
    st.rewards.copy_(0.1 * r(T, N, 1)); st.dones.copy_((torch.rand(T, N, 1, generator=g) < 0.1).byte())
    st.values.copy_(r(T, N, 1)); st.mu.copy_(st.actions + 0.3 * r(T, N, na)); st.sigma.fill_(1.0)
    st.actions_log_prob.copy_((-0.5 * (st.actions - st.mu) ** 2 - 0.9189385).sum(-1, keepdim=True))
    st.env_bins.zero_()
    st.step = T
    last_values = r(N, 1)
    rec = {"in_" + k: getattr(st, k).clone() for k in ("observations", "privileged_observations", "observation_histories",
                                                        "actions", "rewards", "dones", "values", "mu", "sigma", "actions_log_prob")}
    st.compute_returns(last_values, PPO_Args.gamma, PPO_Args.lam)
    rec["out_returns"], rec["out_advantages"] = st.returns.clone(), st.advantages.clone()
    torch.manual_seed(seed + 2)
    losses = alg.update()
    out = {"init_" + k: v for k, v in init.items()}
    out.update({"final_" + k: v.clone() for k, v in ac.state_dict().items()})
    out.update(rec)
    out.update(last_values=last_values, losses=torch.tensor(losses), final_lr=torch.tensor(alg.learning_rate),
               dims=torch.tensor([N, T, no, npv, H, na]), seed=torch.tensor(seed))
    import json
    np.savez_compressed(os.path.join(HERE, name), **flat(out), ppo_args=np.array(json.dumps(over or {})))
    print(name, "losses", losses, "lr", alg.learning_rate)


def gen_ppo_rma(seed=4):
    """reference go1_gym_learn.ppo (the older teacher-student runner): compute_returns + PPO.update on a fixed rollout."""
    ml = types.ModuleType("ml_logger")
    ml.logger = object()
    sys.modules["ml_logger"] = ml
    from go1_gym_learn.ppo.actor_critic import ActorCritic, AC_Args
    from go1_gym_learn.ppo.ppo import PPO, PPO_Args
    AC_Args.actor_hidden_dims = [32, 16]
    AC_Args.critic_hidden_dims = [24, 16]
    AC_Args.adaptation_module_branch_hidden_dims = [[16, 8]]
    AC_Args.env_factor_encoder_branch_input_dims = [5]
    AC_Args.env_factor_encoder_branch_latent_dims = [4]
    AC_Args.env_factor_encoder_branch_hidden_dims = [[12, 8]]
    N, T, no, npv, H, na = 20, 6, 10, 5, 3, 12
    torch.manual_seed(seed)
    ac = ActorCritic(no, npv, no * H, na)
    init = {k: v.clone() for k, v in ac.state_dict().items()}
    alg = PPO(ac, device="cpu")
    alg.init_storage(N, T, [no], [npv], [no * H], [na])
    st = alg.storage
    g = torch.Generator().manual_seed(seed + 1)
    r = lambda *s: torch.randn(*s, generator=g)
    st.observations.copy_(r(T, N, no)); st.privileged_observations.copy_(r(T, N, npv))
    st.observation_histories.copy_(r(T, N, no * H)); st.actions.copy_(r(T, N, na))
    st.rewards.copy_(0.1 * r(T, N, 1)); st.dones.copy_((torch.rand(T, N, 1, generator=g) < 0.1).byte())
    st.values.copy_(r(T, N, 1)); st.mu.copy_(st.actions + 0.3 * r(T, N, na)); st.sigma.fill_(1.0)
    st.actions_log_prob.copy_((-0.5 * (st.actions - st.mu) ** 2 - 0.9189385).sum(-1, keepdim=True))
    st.env_bins.zero_()
    st.step = T
    last_values = r(N, 1)
    rec = {"in_" + k: getattr(st, k).clone() for k in ("observations", "privileged_observations", "observation_histories",
                                                        "actions", "rewards", "dones", "values", "mu", "sigma", "actions_log_prob")}
    st.compute_returns(last_values, PPO_Args.gamma, PPO_Args.lam)
    rec["out_returns"], rec["out_advantages"] = st.returns.clone(), st.advantages.clone()
    # the rollout surface on fixed inputs: teacher / student means and the value (deterministic parts of act())
    probe_obs, probe_priv, probe_hist = r(7, no), r(7, npv), r(7, no * H)
    with torch.no_grad():
        rec["probe_obs"], rec["probe_priv"], rec["probe_hist"] = probe_obs, probe_priv, probe_hist
        rec["probe_teacher"] = ac.act_teacher(probe_obs, probe_priv)
        rec["probe_student"] = ac.act_student(probe_obs, probe_hist)
        rec["probe_value"] = ac.evaluate(probe_obs, probe_priv)
    torch.manual_seed(seed + 2)
    losses = alg.update()
    out = {"init_" + k: v for k, v in init.items()}
    out.update({"final_" + k: v.clone() for k, v in ac.state_dict().items()})
    out.update(rec)
    out.update(last_values=last_values, losses=torch.tensor(losses), final_lr=torch.tensor(alg.learning_rate),
               dims=torch.tensor([N, T, no, npv, H, na]), seed=torch.tensor(seed))
    np.savez_compressed(os.path.join(HERE, "ppo_rma.npz"), **flat(out))
    print("ppo (teacher-student runner) losses", losses, "lr", alg.learning_rate)


# ---- Philox4x32-10 (Random123) in numpy: the uniforms the oracle / kernel draw for an env (go1_math.h rng_uniform) -----
def philox_uniform(seed, env, step, purpose, idx):
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    c = [env & 0xFFFFFFFF, step & 0xFFFFFFFF, purpose, idx >> 2]
    k = [seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF]
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & 0xFFFFFFFF, (p0 >> 32) ^ c[3] ^ k[1], p0 & 0xFFFFFFFF]
        k = [(k[0] + W0) & 0xFFFFFFFF, (k[1] + W1) & 0xFFFFFFFF]
    return np.float32(c[idx & 3] >> 8) * np.float32(1.0 / 16777216.0)


RESAMPLE_MODES = {"gaitwise": dict(gaitwise_curricula=True, exclusive_phase_offset=False, balance_gait_distribution=False, binary_phases=True),
                  "exclusive": dict(gaitwise_curricula=False, exclusive_phase_offset=True, balance_gait_distribution=False, binary_phases=True),
                  "balance": dict(gaitwise_curricula=False, exclusive_phase_offset=False, balance_gait_distribution=True, binary_phases=False),
                  "plain": dict(gaitwise_curricula=False, exclusive_phase_offset=False, balance_gait_distribution=False, binary_phases=False),
                  # the same four branches with the other setting of `binary_phases` (legged_robot.py:814-817, 1361)
                  "gaitwise_smooth": dict(gaitwise_curricula=True, exclusive_phase_offset=False, balance_gait_distribution=False, binary_phases=False),
                  "exclusive_smooth": dict(gaitwise_curricula=False, exclusive_phase_offset=True, balance_gait_distribution=False, binary_phases=False),
                  "balance_binary": dict(gaitwise_curricula=False, exclusive_phase_offset=False, balance_gait_distribution=True, binary_phases=True),
                  "plain_binary": dict(gaitwise_curricula=False, exclusive_phase_offset=False, balance_gait_distribution=False, binary_phases=True)}


def gen_resample(mode, N=96, seed=31, sim_seed=12345, step=77, purpose=6):
    """reference LeggedRobot._resample_commands + RewardThresholdCurriculum.update (legged_robot.py:710-824,
    curriculum.py:135-154) on a mock env.  The two RNG consumers are fed the uniforms the oracle / kernel draw for the
    same (seed, env, step, purpose): `torch.rand` returns column 0 (category) and, for the phase-exclusion branches,
    column 17; `curriculum.sample` is replaced by the inverse-CDF draw from column 1 (numpy's `choice(p=...)` is the same
    searchsorted(cdf, u, 'right')) over the weights in force BEFORE this step's update plus the in-cell
    jitter from columns 2..16.  Everything else — success test, weight update with neighbourhoods, category assignment,
    gait remaps, binary phases, small-command zeroing, sum reset — is the reference's own code."""
    np.int = int                                                   # legged_robot.py:1362 uses the removed alias (App. D13)
    e, LR = make_env("train", N, seed)
    c = e.cfg.commands
    for k, v in RESAMPLE_MODES[mode].items():
        setattr(c, k, v)
    LR._init_command_distribution(e, torch.arange(N))
    rng = np.random.default_rng(seed + 1)
    ncat, nb = len(e.category_names), len(e.curricula[0])
    for cur in e.curricula:                                         # a frontier: some bins partly open, some closed
        on = np.flatnonzero(cur.weights > 0)
        cur.weights[rng.choice(on, len(on) // 3, replace=False)] = 0.4
        off = np.flatnonzero(cur.weights == 0)
        cur.weights[rng.choice(off, 12, replace=False)] = 0.2
    w0 = np.stack([cur.weights.copy() for cur in e.curricula])
    open_bins = [np.flatnonzero(w > 0) for w in w0]
    e.env_command_categories = rng.integers(0, ncat, N)
    e.env_command_bins = np.array([rng.choice(open_bins[k]) for k in e.env_command_categories])
    env_ids = torch.tensor(np.sort(rng.choice(N, (2 * N) // 3, replace=False)), dtype=torch.long)
    ep_len = min(int(e.cfg.env.max_episode_length), int(c.resampling_time / e.dt))
    for key in ("tracking_lin_vel", "tracking_ang_vel", "tracking_contacts_shaped_force", "tracking_contacts_shaped_vel"):
        thr = e.curriculum_thresholds[key] * e.reward_scales[key]
        mult = rng.choice([0.6, 1.5], N, p=[0.25, 0.75])            # clear of the threshold either way
        e.command_sums[key] = torch.tensor(ep_len * thr * mult, dtype=torch.float)
    cmd0 = e.commands.clone()
    sums0 = torch.stack([e.command_sums[k] for k in e.command_sums]).clone()
    bins0, cats0 = np.array(e.env_command_bins).copy(), np.array(e.env_command_categories).copy()
    U = np.array([[philox_uniform(sim_seed, int(i), step, purpose, j) for j in range(18)] for i in range(N)], dtype=np.float32)
    lim = [c.limit_vel_x, c.limit_vel_y, c.limit_vel_yaw, c.limit_body_height, c.limit_gait_frequency, c.limit_gait_phase,
           c.limit_gait_offset, c.limit_gait_bound, c.limit_gait_duration, c.limit_footswing_height, c.limit_body_pitch,
           c.limit_body_roll, c.limit_stance_width, c.limit_stance_length, c.limit_aux_reward_coef]
    nbins = [c.num_bins_vel_x, c.num_bins_vel_y, c.num_bins_vel_yaw, c.num_bins_body_height, c.num_bins_gait_frequency,
             c.num_bins_gait_phase, c.num_bins_gait_offset, c.num_bins_gait_bound, c.num_bins_gait_duration,
             c.num_bins_footswing_height, c.num_bins_body_pitch, c.num_bins_body_roll, c.num_bins_stance_width,
             c.num_bins_stance_length, c.num_bins_aux_reward_coef]
    f32 = np.float32
    cdf_pre = []
    for w in w0:
        cs = np.cumsum(w.astype(np.float32).astype(np.float64))
        cdf_pre.append((cs / cs[-1]).astype(np.float32))
    floats = torch.tensor(U[env_ids.numpy(), 0])
    per = 1.0 / ncat
    cat_ids = [env_ids[torch.logical_and(per * i <= floats, floats < per * (i + 1))] for i in range(ncat)]

    def make_sample(i, cur):
        def sample(batch_size, low=None, high=None):
            ids = cat_ids[i].numpy()
            assert len(ids) == batch_size
            out, inds = np.zeros((batch_size, 15)), np.zeros(batch_size, dtype=int)
            for r, env in enumerate(ids):
                b = int(np.searchsorted(cdf_pre[i], U[env, 1], side="right"))
                b = min(b, nb - 1)
                inds[r] = b
                rem = b
                for kx in range(14, -1, -1):
                    idx = rem % nbins[kx]
                    rem //= nbins[kx]
                    bs = (f32(lim[kx][1]) - f32(lim[kx][0])) / f32(nbins[kx])
                    centroid = f32(lim[kx][0]) + bs * (f32(idx) + f32(0.5))
                    assert abs(float(centroid) - cur.grid[kx, b]) < 1e-5      # same bin -> centroid decode as the reference grid
                    out[r, kx] = centroid + (U[env, 2 + kx] - f32(0.5)) * bs
            return out, inds
        return sample
    for i, cur in enumerate(e.curricula):
        cur.sample = make_sample(i, cur)
    calls = [torch.tensor(U[env_ids.numpy(), 0]), torch.tensor(U[env_ids.numpy(), 17])]
    real_rand = torch.rand
    torch.rand = lambda *a, **k: calls.pop(0)
    try:
        LR._resample_commands(e, env_ids)
    finally:
        torch.rand = real_rand
    out = dict(env_ids=env_ids.numpy(), weights0=w0, weights1=np.stack([cur.weights for cur in e.curricula]),
               commands0=cmd0.numpy(), commands1=e.commands.numpy(), command_sums0=sums0.numpy(),
               command_sums1=torch.stack([e.command_sums[k] for k in e.command_sums]).numpy(),
               bins0=bins0, cats0=cats0, bins1=np.asarray(e.env_command_bins), cats1=np.asarray(e.env_command_categories),
               uniforms=U, sim_seed=np.array(sim_seed), step=np.array(step), purpose=np.array(purpose), ep_len=np.array(ep_len),
               command_sum_names=np.array(list(e.command_sums)), category_names=np.array(e.category_names))
    np.savez_compressed(os.path.join(HERE, f"resample_{mode}.npz"), **out)
    ch = np.abs(out["weights1"] - w0).sum()
    print("resample", mode, "envs", len(env_ids), "weight mass added", float(ch), "zeroed xy", int((np.abs(out["commands1"][env_ids.numpy(), :2]).sum(1) == 0).sum()))


def gen_reset(N=80, seed=41, sim_seed=4242, step=311):
    """reference `_randomize_dof_props`, `_reset_dofs`, `_reset_root_states` (legged_robot.py:645-665,948-1001) on a mock env,
    with `torch.rand` / `torch_rand_float` serving the Philox uniforms the oracle / kernel draw for the same
    (seed, env, step): purpose 4 (DOF properties at reset) columns 0..14, purpose 2 (reset) columns 0..20."""
    import go1_gym.envs.base.legged_robot as ref_mod
    e, LR = make_env("alt", N, seed)                               # alt: Kp / Kd factors randomised too
    sys.modules["isaacgym.gymtorch"].unwrap_tensor = lambda t: t
    ref_mod.gymtorch.unwrap_tensor = lambda t: t

    class AnyCall:
        def __getattr__(self, name):
            return lambda *a, **k: True
    e.gym, e.sim, e.dof_state = AnyCall(), None, torch.zeros(N * 12, 2)
    e.eval_cfg = None
    e.cfg.env.record_video = False
    rng = np.random.default_rng(seed + 1)
    e.custom_origins = True
    e.env_origins = torch.tensor(rng.uniform(-20, 20, (N, 3)), dtype=torch.float)
    e.env_origins[:, 2] = torch.tensor(rng.uniform(0, 0.3, N), dtype=torch.float)
    st = e.cfg.init_state
    e.base_init_state = torch.tensor(list(st.pos) + list(st.rot) + list(st.lin_vel) + list(st.ang_vel), dtype=torch.float)
    ids = torch.tensor(np.sort(rng.choice(N, N // 2, replace=False)), dtype=torch.long)
    U_dof = np.array([[philox_uniform(sim_seed, int(i), step, 4, j) for j in range(16)] for i in range(N)], dtype=np.float32)
    U_rst = np.array([[philox_uniform(sim_seed, int(i), step, 2, j) for j in range(24)] for i in range(N)], dtype=np.float32)
    pre = dict(dof_pos0=e.dof_pos.clone(), dof_vel0=e.dof_vel.clone(), root_states0=e.root_states.clone(),
               motor_strengths0=e.motor_strengths.clone(), motor_offsets0=e.motor_offsets.clone(), Kp_factors0=e.Kp_factors.clone(),
               Kd_factors0=e.Kd_factors.clone(), env_origins=e.env_origins.clone())
    idn = ids.numpy()
    q_dof = [torch.tensor(U_dof[idn, 0]), torch.tensor(U_dof[idn, 1:13]), torch.tensor(U_dof[idn, 13]), torch.tensor(U_dof[idn, 14])]
    q_rst = [U_rst[idn, 0:12], U_rst[idn, 12:13], U_rst[idn, 13:14], U_rst[idn, 14:15], U_rst[idn, 15:21]]
    real_rand, real_trf = torch.rand, ref_mod.torch_rand_float
    torch.rand = lambda *a, **k: q_dof.pop(0)

    def trf(lo, hi, shape, device=None):
        u = torch.tensor(q_rst.pop(0))
        assert tuple(u.shape) == tuple(shape)
        return (hi - lo) * u + lo                                   # isaacgym.torch_utils.torch_rand_float
    ref_mod.torch_rand_float = trf
    try:
        LR._randomize_dof_props(e, ids, e.cfg)
        LR._reset_dofs(e, ids, e.cfg)
        LR._reset_root_states(e, ids, e.cfg)
    finally:
        torch.rand, ref_mod.torch_rand_float = real_rand, real_trf
    assert not q_dof and not q_rst
    out = dict(env_ids=idn, dof_pos1=e.dof_pos, dof_vel1=e.dof_vel, root_states1=e.root_states, motor_strengths1=e.motor_strengths,
               motor_offsets1=e.motor_offsets, Kp_factors1=e.Kp_factors, Kd_factors1=e.Kd_factors,
               sim_seed=np.array(sim_seed), step=np.array(step))
    np.savez_compressed(os.path.join(HERE, "reset.npz"), **flat(pre), **flat(out))
    print("reset: envs", len(idn), "yaw span", float(e.root_states[ids, 5].abs().max()))


EVAL_OVERRIDES = {"domain_rand": dict(motor_strength_range=[1.3, 1.5], motor_offset_range=[0.05, 0.08], Kp_factor_range=[1.4, 1.6],
                                      Kd_factor_range=[0.2, 0.4]),
                  "terrain": dict(x_init_range=0.2, y_init_range=0.3, yaw_init_range=0.4, x_init_offset=1.5, y_init_offset=-2.5)}


def gen_reset_eval(N=64, NT=32, seed=43, sim_seed=777, step=123):
    """the train / evaluation dispatch of the reference (`_call_train_eval`, legged_robot.py:531-544) around
    `_randomize_dof_props`, `_reset_dofs`, `_reset_root_states`: environments >= NT are handled with `eval_cfg` (other
    domain-randomisation ranges, other reset distribution).  Same Philox uniforms as gen_reset; every dispatched call
    serves the training ids first, then the evaluation ids (the order `_call_train_eval` calls `func` in)."""
    import copy
    import types
    import go1_gym.envs.base.legged_robot as ref_mod
    e, LR = make_env("alt", N, seed)
    sys.modules["isaacgym.gymtorch"].unwrap_tensor = lambda t: t
    ref_mod.gymtorch.unwrap_tensor = lambda t: t

    class AnyCall:
        def __getattr__(self, name):
            return lambda *a, **k: True
    e.gym, e.sim, e.dof_state = AnyCall(), None, torch.zeros(N * 12, 2)
    e.num_train_envs = NT
    e.cfg.env.record_video = False
    ev = types.SimpleNamespace()
    for sec in ("domain_rand", "terrain", "env"):
        src = getattr(e.cfg, sec)
        ns = types.SimpleNamespace(**{k: copy.deepcopy(getattr(src, k)) for k in dir(src) if not k.startswith("_") and not callable(getattr(src, k))})
        for k, v in EVAL_OVERRIDES.get(sec, {}).items():
            setattr(ns, k, v)
        setattr(ev, sec, ns)
    e.eval_cfg = ev
    rng = np.random.default_rng(seed + 1)
    e.custom_origins = True
    e.env_origins = torch.tensor(rng.uniform(-20, 20, (N, 3)), dtype=torch.float)
    e.env_origins[:, 2] = torch.tensor(rng.uniform(0, 0.3, N), dtype=torch.float)
    st = e.cfg.init_state
    e.base_init_state = torch.tensor(list(st.pos) + list(st.rot) + list(st.lin_vel) + list(st.ang_vel), dtype=torch.float)
    ids = torch.tensor(np.sort(rng.choice(N, N // 2, replace=False)), dtype=torch.long)
    U_dof = np.array([[philox_uniform(sim_seed, int(i), step, 4, j) for j in range(16)] for i in range(N)], dtype=np.float32)
    U_rst = np.array([[philox_uniform(sim_seed, int(i), step, 2, j) for j in range(24)] for i in range(N)], dtype=np.float32)
    pre = dict(dof_pos0=e.dof_pos.clone(), dof_vel0=e.dof_vel.clone(), root_states0=e.root_states.clone(),
               motor_strengths0=e.motor_strengths.clone(), motor_offsets0=e.motor_offsets.clone(), Kp_factors0=e.Kp_factors.clone(),
               Kd_factors0=e.Kd_factors.clone(), env_origins=e.env_origins.clone())
    idn = ids.numpy()
    groups = [idn[idn < NT], idn[idn >= NT]]
    assert len(groups[0]) > 4 and len(groups[1]) > 4
    q_dof, q_dofs, q_root = [], [], []
    for gidx in groups:
        q_dof += [torch.tensor(U_dof[gidx, 0]), torch.tensor(U_dof[gidx, 1:13]), torch.tensor(U_dof[gidx, 13]), torch.tensor(U_dof[gidx, 14])]
        q_dofs += [U_rst[gidx, 0:12]]
        q_root += [U_rst[gidx, 12:13], U_rst[gidx, 13:14], U_rst[gidx, 14:15], U_rst[gidx, 15:21]]
    q_rst = q_dofs + q_root
    real_rand, real_trf = torch.rand, ref_mod.torch_rand_float
    torch.rand = lambda *a, **k: q_dof.pop(0)

    def trf(lo, hi, shape, device=None):
        u = torch.tensor(q_rst.pop(0))
        assert tuple(u.shape) == tuple(shape)
        return (hi - lo) * u + lo
    ref_mod.torch_rand_float = trf
    try:
        LR._call_train_eval(e, lambda i_, c_: LR._randomize_dof_props(e, i_, c_), ids)
        LR._call_train_eval(e, lambda i_, c_: LR._reset_dofs(e, i_, c_), ids)
        LR._call_train_eval(e, lambda i_, c_: LR._reset_root_states(e, i_, c_), ids)
    finally:
        torch.rand, ref_mod.torch_rand_float = real_rand, real_trf
    assert not q_dof and not q_rst
    out = dict(env_ids=idn, dof_pos1=e.dof_pos, dof_vel1=e.dof_vel, root_states1=e.root_states, motor_strengths1=e.motor_strengths,
               motor_offsets1=e.motor_offsets, Kp_factors1=e.Kp_factors, Kd_factors1=e.Kd_factors,
               sim_seed=np.array(sim_seed), step=np.array(step), num_train=np.array(NT))
    np.savez_compressed(os.path.join(HERE, "reset_eval.npz"), **flat(pre), **flat(out))
    ms = e.motor_strengths[ids]
    print("reset_eval: train ids", len(groups[0]), "eval ids", len(groups[1]), "eval strength min", float(ms[len(groups[0]):].min()))


def gen_gravity(seed=61, sim_seed=4242):
    """the gravity impulse schedule (legged_robot.py:546-561 `_randomize_gravity`, :701-705 its cadence, :1549 the draw at
    creation): the reference's own statements, extracted from `_post_physics_step_callback` with inspect and executed for
    common_step_counter = 1 .. 3 intervals on a mock env (variant "dr"); `torch.rand(3)` serves the Philox uniforms the
    oracle / kernel draw for (seed, all-envs, epoch = counter // interval, purpose 8).  Saved: the gravity vector in force
    after the callback of every counter value (index 0 = after creation)."""
    import inspect
    import textwrap
    import go1_gym.envs.base.legged_robot as ref_mod
    e, LR = make_env("dr", 4, seed, mild=True)
    ref_mod.gymapi.Vec3 = lambda x, y, z: (float(x), float(y), float(z))
    params = Mock()

    class Gym:
        def get_sim_params(self, sim):
            return params

        def set_sim_params(self, sim, p):
            pass
    e.gym, e.sim = Gym(), None
    e._randomize_gravity = lambda *a, **k: LR._randomize_gravity(e, *a, **k)
    dr = e.cfg.domain_rand
    interval, duration = int(dr.gravity_rand_interval), int(dr.gravity_rand_duration)
    lines = inspect.getsource(LR._post_physics_step_callback).splitlines()
    first = next(i for i, l in enumerate(lines) if "gravity_rand_interval" in l)
    last = max(i for i, l in enumerate(lines) if "_randomize_gravity(" in l)
    block = compile(textwrap.dedent("\n".join(lines[first:last + 1])), "<reference legged_robot.py gravity cadence>", "exec")
    real_rand = torch.rand
    torch.rand = lambda *a, **k: torch.tensor([philox_uniform(sim_seed, 0xFFFFFFFF, e.common_step_counter // interval, 8, i) for i in range(3)])
    T = 3 * interval + 5
    g = np.zeros((T, 3), np.float32)
    gv = np.zeros((T, 3), np.float32)
    try:
        e.common_step_counter = 0
        LR._randomize_gravity(e)                                   # create_envs (:1549)
        g[0], gv[0] = params.gravity, e.gravity_vec[0].numpy()
        for c in range(1, T):
            e.common_step_counter = c
            exec(block, {"self": e, "torch": torch, "int": int})
            g[c], gv[c] = params.gravity, e.gravity_vec[0].numpy()
    finally:
        torch.rand = real_rand
    np.savez_compressed(os.path.join(HERE, "gravity.npz"), gravity=g, gravity_vec=gv, interval=np.array(interval),
                        duration=np.array(duration), sim_seed=np.array(sim_seed))
    print("gravity: interval", interval, "duration", duration, "impulse steps", int((np.abs(g[:, :2]).max(1) > 0).sum()), "of", T)


TERRAIN_LAYOUT = dict(
    train=dict(mesh_type="trimesh", num_rows=3, num_cols=5, terrain_length=4.0, terrain_width=4.0, border_size=2.0,
               horizontal_scale=0.1, vertical_scale=0.005, curriculum=True, selected=False, difficulty_scale=1.0,
               terrain_proportions=[0.1, 0.1, 0.2, 0.2, 0.2, 0.0, 0.0, 0.0, 0.2], terrain_noise_magnitude=0.03,
               terrain_smoothness=0.005, max_platform_height=0.2, slope_treshold=0.75),
    eval=dict(mesh_type="trimesh", num_rows=2, num_cols=7, terrain_length=4.0, terrain_width=4.0, border_size=2.0,
              horizontal_scale=0.1, vertical_scale=0.005, curriculum=False, selected=False, difficulty_scale=1.0,
              terrain_proportions=[0.2, 0.2, 0.1, 0.1, 0.2, 0.0, 0.0, 0.0, 0.1, 0.1], terrain_noise_magnitude=0.02,
              terrain_smoothness=0.005, max_platform_height=0.2, slope_treshold=0.75))


def gen_terrain_layout(seed=71):
    """reference go1_gym/utils/terrain.py `Terrain` (tile grid, borders, the evaluation region appended behind the training one
    :37-54, `make_terrain`'s choice / difficulty -> generator mapping :114-159, env origins :161-179) executed with this repo's
    sub-terrain generators standing in for the closed `isaacgym.terrain_utils` (so the fixture pins the LAYOUT code; the
    generators themselves are re-derived) — training grid in curriculum mode, evaluation grid in randomised mode."""
    gens = load_private("_wtw_terrain", os.path.join(PKG, "go1_gym", "utils", "terrain.py"))
    tu = sys.modules["isaacgym.terrain_utils"]
    for n in ("SubTerrain", "random_uniform_terrain", "pyramid_sloped_terrain", "pyramid_stairs_terrain", "discrete_obstacles_terrain",
              "stepping_stones_terrain"):
        setattr(tu, n, getattr(gens, n))
    tu.convert_heightfield_to_trimesh = lambda *a, **k: (None, None)
    import isaacgym
    isaacgym.terrain_utils = tu
    from go1_gym.utils.terrain import Terrain                     # the REFERENCE class
    out = {}
    for tag, with_eval in (("solo", False), ("split", True)):
        tr, ev = types.SimpleNamespace(**TERRAIN_LAYOUT["train"]), types.SimpleNamespace(**TERRAIN_LAYOUT["eval"])
        np.random.seed(seed)
        T = Terrain(tr, 32, ev, 16) if with_eval else Terrain(tr, 32)
        out[f"{tag}_heights"] = T.height_field_raw.copy()
        out[f"{tag}_train_origins"] = tr.env_origins.copy()
        out[f"{tag}_tot"] = np.array([T.tot_rows, T.tot_cols])
        if with_eval:
            out[f"{tag}_eval_origins"] = ev.env_origins.copy()
            out[f"{tag}_eval_offsets"] = np.array([ev.x_offset, ev.rows_offset])
    import json
    np.savez_compressed(os.path.join(HERE, "terrain_layout.npz"), seed=np.array(seed), config=np.array(json.dumps(TERRAIN_LAYOUT)), **out)
    print("terrain_layout:", out["solo_tot"], out["split_tot"], "non-flat samples", int((out["split_heights"] != 0).sum()))


def gen_pretrain_parameters():
    """the `parameters.pkl` the reference ships with its pretrained run (runs/gait-conditioned-agility/pretrain-v0/train/
    025417.456545; written by `logger.log_params(..., Cfg=vars(Cfg))`, train.py:209-210, read back by scripts/play.py:37-47) as
    JSON: the run's effective configuration INCLUDING the values the reference derived at construction (`_parse_cfg`:
    max_episode_length, push / rand / gravity intervals; `Terrain`: tile grid sizes, env origins)."""
    import io
    import json
    import pickle

    class CpuUnpickler(pickle.Unpickler):                         # (the tensors were pickled on a CUDA device)
        def find_class(self, module, name):
            if module == "torch.storage" and name == "_load_from_bytes":
                return lambda b: torch.load(io.BytesIO(b), map_location="cpu", weights_only=False)
            return super().find_class(module, name)

    def plain(v):
        if isinstance(v, dict):
            return {str(k): plain(x) for k, x in v.items()}
        if isinstance(v, (list, tuple)):
            return [plain(x) for x in v]
        if isinstance(v, torch.Tensor):
            return v.tolist()
        if isinstance(v, np.ndarray):
            return v.tolist()
        if isinstance(v, np.generic):
            return v.item()
        return v
    with open(os.path.join(REF, "runs/gait-conditioned-agility/pretrain-v0/train/025417.456545/parameters.pkl"), "rb") as f:
        d = CpuUnpickler(f).load()
    out = plain(d)
    for k in ("row_indices", "col_indices"):                      # arange(tot_rows): 1500 integers each, implied by tot_rows / tot_cols
        out["Cfg"]["terrain"].pop(k, None)
    with open(os.path.join(HERE, "pretrain_parameters.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("pretrain_parameters:", list(out), "max_episode_length", out["Cfg"]["env"]["max_episode_length"])


def gen_traj_utils(seed=81):
    """reference go1_gym_learn/utils/utils.py `split_and_pad_trajectories` / `unpad_trajectories` (:5-43) on a random rollout."""
    from go1_gym_learn.utils.utils import split_and_pad_trajectories, unpad_trajectories        # the REFERENCE functions
    g = torch.Generator().manual_seed(seed)
    T, N, D = 24, 7, 5
    x = torch.randn(T, N, D, generator=g)
    dones = (torch.rand(T, N, 1, generator=g) < 0.15).to(torch.uint8)
    dones[:, 3] = 0                                               # one environment that never ends: a full-length piece
    padded, masks = split_and_pad_trajectories(x, dones)
    back = unpad_trajectories(padded, masks)
    np.savez_compressed(os.path.join(HERE, "traj_utils.npz"), x=x.numpy(), dones=dones.numpy(), padded=padded.numpy(),
                        masks=masks.numpy(), back=back.numpy())
    print("traj_utils: pieces", padded.shape[1], "longest", padded.shape[0])


def gen_sum_curriculum(seed=91):
    """reference curriculum.py `SumCurriculum` (:92-109) on random updates (the module imports standalone)."""
    spec = importlib.util.spec_from_file_location("_ref_curriculum", os.path.join(REF, "go1_gym/envs/base/curriculum.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules.setdefault("matplotlib", types.ModuleType("matplotlib"))
    sys.modules.setdefault("matplotlib.pyplot", types.ModuleType("matplotlib.pyplot"))
    sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(seed)
    c = mod.SumCurriculum(seed=3, x=(-1.0, 1.0, 5), y=(0.0, 2.0, 3), z=(0.0, 1.0, 2))
    bins, errs = [], []
    for _ in range(6):
        b = rng.integers(0, len(c), 12)
        e = rng.uniform(0, 1, 12)
        c.update(b, e, 0.4)
        bins.append(b)
        errs.append(e)
    np.savez_compressed(os.path.join(HERE, "sum_curriculum.npz"), bins=np.array(bins), errs=np.array(errs), success=c.success,
                        trials=c.trials, rates_all=c.success_rates("x", "y", "z"), rates_x=c.success_rates("x"),
                        rates_xz=c.success_rates("x", "z"), met=np.array([mod.is_met(2.0, 0.5, 0.3), mod.is_met(2.0, 0.7, 0.3),
                                                                          mod.key_is_met(None, None, 10, "k", 0, 0.1)]))
    print("sum_curriculum: trials", int(c.trials.sum()), "successes", int(c.success.sum()))


def gen_recurrent_batches(seed=95):
    """reference `RolloutStorage.reccurent_mini_batch_generator` (rollout_storage.py:141-180) on a random filled storage."""
    from go1_gym_learn.ppo_cse.rollout_storage import RolloutStorage            # the REFERENCE class
    g = torch.Generator().manual_seed(seed)
    T, N = 12, 6
    st = RolloutStorage(N, T, [7], [2], [21], [3], device="cpu")
    fill = {}
    for name in ("observations", "privileged_observations", "observation_histories", "actions", "values", "advantages", "returns",
                 "actions_log_prob", "mu", "sigma"):
        t = getattr(st, name)
        t.copy_(torch.randn(t.shape, generator=g))
        fill[name] = t.numpy().copy()
    st.dones.copy_((torch.rand(T, N, 1, generator=g) < 0.2).to(torch.uint8))
    fill["dones"] = st.dones.numpy().copy()
    out = {}
    for i, batch in enumerate(st.reccurent_mini_batch_generator(2, num_epochs=2)):
        for j, t in enumerate(batch):
            out[f"b{i}_{j}"] = t.numpy()
    np.savez_compressed(os.path.join(HERE, "recurrent_batches.npz"), **{"in_" + k: v for k, v in fill.items()}, **out)
    print("recurrent_batches:", i + 1, "batches of", len(batch), "tensors")


def gen_cfg_defaults():
    """every default of the reference's configuration classes — `Cfg` as declared (legged_robot_config.py), `Cfg` after
    `config_go1` (go1_config.py), and the `AC_Args` / `PPO_Args` / `RunnerArgs` of both learners — as one JSON tree."""
    import inspect
    import json
    from go1_gym.envs.base.legged_robot_config import Cfg                      # the REFERENCE classes
    from go1_gym.envs.go1.go1_config import config_go1

    def tree(node):
        out = {}
        for k in dir(node):
            if k.startswith("_"):
                continue
            v = getattr(node, k)
            if inspect.isclass(v):
                out[k] = tree(v)
            elif callable(v):
                continue
            elif isinstance(v, tuple):
                out[k] = list(v)
            else:
                out[k] = v
        return out
    res = {"Cfg": tree(Cfg)}
    config_go1(Cfg)
    res["Cfg_go1"] = tree(Cfg)
    for pkg in ("ppo", "ppo_cse"):
        ac = importlib.import_module(f"go1_gym_learn.{pkg}.actor_critic")
        pp = importlib.import_module(f"go1_gym_learn.{pkg}.ppo")
        rn = importlib.import_module(f"go1_gym_learn.{pkg}")
        res[pkg] = {"AC_Args": tree(ac.AC_Args), "PPO_Args": tree(pp.PPO_Args), "RunnerArgs": tree(rn.RunnerArgs)}
    with open(os.path.join(HERE, "cfg_defaults.json"), "w") as f:
        json.dump(res, f, indent=0, sort_keys=True)
    print("cfg_defaults: sections", len(res["Cfg"]), "leaves", sum(len(v) if isinstance(v, dict) else 1 for v in res["Cfg"].values()))


HISTORY_SCRIPT = ["reset", "get_observations", "step", "step", "get_observations", "step", "step", "step", "step", "step", "step",
                  "get_observations", "reset", "step", "step"]           # (`reset_idx` of the wrapper cannot run: gym.Wrapper has none)


def gen_history_trace():
    """reference `HistoryWrapper` (history_wrapper.py:6-41) driven through HISTORY_SCRIPT over a mock environment whose k-th
    observation is the constant k: which observation sits in which history slot after every call (0 = still empty) — the
    extra shift of `get_observations` (:29), the zeroing of `reset` / `reset_idx` (:32-41)."""
    import json

    class Wrapper:                                                 # gym.Wrapper's two behaviours the class relies on
        def __init__(self, env):
            self.env = env

        def __getattr__(self, name):
            return getattr(self.__dict__["env"], name)

        def step(self, action):
            return self.env.step(action)

        def reset(self, **kwargs):
            return self.env.reset(**kwargs)
    sys.modules["gym"].Wrapper = Wrapper
    mod = load_private("_ref_history_wrapper", os.path.join(REF, "go1_gym/envs/wrappers/history_wrapper.py"))

    class Env:
        num_envs, num_obs, num_privileged_obs, device = 2, 3, 2, "cpu"

        def __init__(self):
            self.cfg = Mock()
            self.cfg.env = Mock()
            self.cfg.env.num_observation_history = 4
            self.k = 0
            self.obs_buf = torch.zeros(2, 3)

        def _new(self):
            self.k += 1
            self.obs_buf = torch.full((2, 3), float(self.k))
            return self.obs_buf

        def step(self, action):
            return self._new(), torch.zeros(2), torch.zeros(2), {"privileged_obs": torch.zeros(2, 2)}

        def reset(self):
            return self._new()

        def reset_idx(self, env_ids):
            return None

        def get_observations(self):
            return self.obs_buf

        def get_privileged_observations(self):
            return torch.zeros(2, 2)
    w = mod.HistoryWrapper(Env())
    trace = []
    for call in HISTORY_SCRIPT:
        if call == "reset":
            out = w.reset()
        elif call == "get_observations":
            out = w.get_observations()
        elif call == "step":
            out = w.step(torch.zeros(2, 12))[0]
        h = out["obs_history"].reshape(2, 4, 3)
        assert bool((h == h[..., :1]).all())
        trace.append(h[..., 0].to(torch.int64).tolist())
    with open(os.path.join(HERE, "history_trace.json"), "w") as f:
        json.dump({"script": HISTORY_SCRIPT, "trace": trace}, f)
    print("history_trace:", trace[-1])


def gen_pretrain_jit_layout():
    """structure of the TorchScript `adaptation_module_latest.jit` the reference ships with its pretrained run (what scripts/
    play.py:17-29 and the deployment stack load): parameter names / shapes and the module sequence — no weights."""
    import json
    m = torch.jit.load(os.path.join(REF, "runs/gait-conditioned-agility/pretrain-v0/train/025417.456545/checkpoints/adaptation_module_latest.jit"),
                       map_location="cpu")
    out = {"original_name": m.original_name, "state_dict": {k: list(v.shape) for k, v in m.state_dict().items()},
           "children": [[n, c.original_name] for n, c in m.named_children()],
           "dtypes": sorted({str(v.dtype) for v in m.state_dict().values()})}
    with open(os.path.join(HERE, "pretrain_jit_layout.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print("pretrain_jit_layout:", out["children"])


def gen_rollout(seed=21):
    """reference go1_gym_learn.ppo_cse rollout path: PPO.act (ppo.py:65-77; sampling from torch's global generator) ->
    process_env_step (:79-91; time-out bootstrap) for T steps -> compute_returns, on fixed observation streams."""
    ml = types.ModuleType("ml_logger")
    ml.logger = object()
    sys.modules["ml_logger"] = ml
    from go1_gym_learn.ppo_cse.actor_critic import ActorCritic, AC_Args
    from go1_gym_learn.ppo_cse.ppo import PPO
    AC_Args.actor_hidden_dims, AC_Args.critic_hidden_dims, AC_Args.adaptation_module_branch_hidden_dims = [32, 16], [24, 16], [16, 8]
    N, T, no, npv, H, na = 9, 5, 10, 2, 3, 12
    torch.manual_seed(seed)
    ac = ActorCritic(no, npv, no * H, na)
    init = {k: v.clone() for k, v in ac.state_dict().items()}
    alg = PPO(ac, device="cpu")
    alg.init_storage(N, T, [no], [npv], [no * H], [na])
    g = torch.Generator().manual_seed(seed + 1)
    r = lambda *s: torch.randn(*s, generator=g)
    obs, priv, hist = r(T + 1, N, no), r(T + 1, N, npv), r(T + 1, N, no * H)
    rew = 0.1 * r(T, N)
    dones = (torch.rand(T, N, generator=g) < 0.2)
    touts = dones & (torch.rand(T, N, generator=g) < 0.5)
    bins = torch.randint(0, 50, (T, N), generator=g).float()
    torch.manual_seed(seed + 2)
    acts = []
    for t in range(T):
        acts.append(alg.act(obs[t], priv[t], hist[t]).clone())
        alg.process_env_step(rew[t].clone(), dones[t].clone(), {"env_bins": bins[t], "time_outs": touts[t]})
    alg.compute_returns(hist[T], priv[T])
    st = alg.storage
    out = {"init_" + k: v for k, v in init.items()}
    out.update({"st_" + k: getattr(st, k).clone() for k in ("observations", "privileged_observations", "observation_histories", "actions",
                                                             "rewards", "dones", "values", "actions_log_prob", "mu", "sigma", "env_bins",
                                                             "returns", "advantages")})
    out.update(obs=obs, priv=priv, hist=hist, rew=rew, dones_in=dones, touts=touts, bins=bins, acts=torch.stack(acts),
               dims=torch.tensor([N, T, no, npv, H, na]), seed=torch.tensor(seed))
    np.savez_compressed(os.path.join(HERE, "rollout.npz"), **flat(out))
    print("rollout: bootstrapped", int(touts.sum()), "of", int(dones.sum()), "dones")


def gen_callbacks(N=64, seed=51, sim_seed=999, step=200, x_offset_px=0, name="callbacks.npz"):
    """the randomising branches of `_post_physics_step_callback` that scripts/train.py leaves off (legged_robot.py:675-708):
    `_teleport_robots` :1028-1051, `_push_robots` :1017-1026, `_randomize_dof_props` :645-665 and
    `_randomize_rigid_body_props` :611-633 on their episode-length cadence — the reference's own methods on a mock env
    (variant "dr"), `torch.rand` / `torch_rand_float` serving the Philox uniforms the oracle / kernel draw for the same
    (seed, env, step): purpose 7 (push) columns 0..1, purpose 3 (DOF properties) 0..14, purpose 9 (rigid properties) 0..5."""
    import go1_gym.envs.base.legged_robot as ref_mod
    e, LR = make_env("dr", N, seed, mild=True)
    sys.modules["isaacgym.gymtorch"].unwrap_tensor = lambda t: t
    ref_mod.gymtorch.unwrap_tensor = lambda t: t

    class AnyCall:
        def __getattr__(self, name):
            return lambda *a, **k: True
    e.gym, e.sim = AnyCall(), None
    e.eval_cfg = None
    e.cfg.env.record_video = False
    cfg = e.cfg
    rng = np.random.default_rng(seed + 1)
    ter, dr = cfg.terrain, cfg.domain_rand
    ter.x_offset = x_offset_px             # set by the reference's Terrain class (terrain.py:42 / :51, in samples); no terrain object on the mock
    span_x, span_y = ter.terrain_length * ter.num_rows, ter.terrain_width * ter.num_cols
    xo = int(ter.x_offset * ter.horizontal_scale)
    kind = rng.integers(0, 5, N)
    e.root_states[:, 0] = torch.tensor(np.where(kind == 0, ter.teleport_thresh + xo - 0.15, np.where(kind == 1, span_x - ter.teleport_thresh + xo + 0.15,
                                                rng.uniform(5.0, span_x - 5.0, N) + xo)), dtype=torch.float)
    kind = rng.integers(0, 5, N)
    e.root_states[:, 1] = torch.tensor(np.where(kind == 0, ter.teleport_thresh - 0.15, np.where(kind == 1, span_y - ter.teleport_thresh + 0.15,
                                                rng.uniform(5.0, span_y - 5.0, N))), dtype=torch.float)
    pi, ri = int(dr.push_interval), int(dr.rand_interval)
    choice = rng.integers(0, 4, N)
    mult = rng.integers(1, 6, N)
    ep = np.where(choice == 0, pi * mult, np.where(choice == 1, ri * mult, np.where(choice == 2, pi * ri, 3 + 17 * mult)))
    e.episode_length_buf = torch.tensor(ep, dtype=torch.long)                 # values AFTER the increment of post_physics_step (:101)
    inp = dict(root_states=e.root_states.clone(), dof_pos=e.dof_pos.clone(), dof_vel=e.dof_vel.clone(),
               gravity=torch.tensor([0.0, 0.0, -9.8]), foot_positions=e.foot_positions.clone(),
               foot_velocities=e.foot_velocities.clone(), prev_foot_velocities=e.prev_foot_velocities.clone(),
               contact_forces=e.contact_forces.clone(), actions=e.actions.clone(), last_actions=e.last_actions.clone(),
               last_last_actions=e.last_last_actions.clone(), joint_pos_target=e.joint_pos_target.clone(),
               last_joint_pos_target=e.last_joint_pos_target.clone(),
               last_last_joint_pos_target=e.last_last_joint_pos_target.clone(), last_dof_vel=e.last_dof_vel.clone(),
               torques=e.torques.clone(), last_contacts=e.last_contacts.clone(), commands=e.commands.clone(),
               gait_indices=e.gait_indices.clone(), episode_length_buf=e.episode_length_buf.clone(),
               friction_coeffs=e.friction_coeffs[:, 0].clone(), restitutions=e.restitutions[:, 0].clone(),
               payloads=e.payloads.clone(), com_displacements=e.com_displacements.clone(),
               motor_strengths=e.motor_strengths.clone(), motor_offsets=e.motor_offsets.clone(),
               Kp_factors=e.Kp_factors.clone(), Kd_factors=e.Kd_factors.clone(),
               episode_sums=torch.stack([e.episode_sums[k] for k in e.episode_sums]).clone(),
               command_sums=torch.stack([e.command_sums[k] for k in e.command_sums]).clone())
    names = dict(episode_sum_names=np.array(list(e.episode_sums)), command_sum_names=np.array(list(e.command_sums)))
    U = {p_: np.array([[philox_uniform(sim_seed, int(i), step, p_, j) for j in range(16)] for i in range(N)], dtype=np.float32) for p_ in (3, 7, 9)}
    all_ids = torch.arange(N)
    push_ids = np.nonzero(ep % pi == 0)[0]
    rand_ids = torch.tensor(np.nonzero(ep % ri == 0)[0], dtype=torch.long)
    rn = rand_ids.numpy()
    q_rand = []                                                                  # in the order the two methods draw
    for flag, cols in ((dr.randomize_motor_strength, slice(0, 1)), (dr.randomize_motor_offset, slice(1, 13)),
                       (dr.randomize_Kp_factor, slice(13, 14)), (dr.randomize_Kd_factor, slice(14, 15))):
        if flag:
            u = U[3][rn, cols]
            q_rand.append(torch.tensor(u[:, 0] if u.shape[1] == 1 else u))
    for flag, cols, squeeze in ((dr.randomize_base_mass, slice(0, 1), True), (dr.randomize_com_displacement, slice(1, 4), False),
                                (dr.randomize_friction, slice(4, 5), False), (dr.randomize_restitution, slice(5, 6), False)):
        if flag:
            u = U[9][rn, cols]
            q_rand.append(torch.tensor(u[:, 0] if squeeze else u))
    q_push = [U[7][push_ids, 0:2]]
    real_rand, real_trf = torch.rand, ref_mod.torch_rand_float
    torch.rand = lambda *a, **k: q_rand.pop(0)

    def trf(lo, hi, shape, device=None):
        u = torch.tensor(q_push.pop(0))
        assert tuple(u.shape) == tuple(shape), (u.shape, shape)
        return (hi - lo) * u + lo
    ref_mod.torch_rand_float = trf
    try:
        LR._teleport_robots(e, all_ids, cfg)
        LR._push_robots(e, all_ids, cfg)
        LR._randomize_dof_props(e, rand_ids, cfg)
        LR._randomize_rigid_body_props(e, rand_ids, cfg)
    finally:
        torch.rand, ref_mod.torch_rand_float = real_rand, real_trf
    assert not q_rand and not q_push, (len(q_rand), len(q_push))
    out = dict(out_root_states=e.root_states, out_motor_strengths=e.motor_strengths, out_motor_offsets=e.motor_offsets,
               out_Kp_factors=e.Kp_factors, out_Kd_factors=e.Kd_factors, out_payloads=e.payloads, out_com_displacements=e.com_displacements,
               out_friction_coeffs=e.friction_coeffs[:, 0], out_restitutions=e.restitutions[:, 0],
               push_ids=push_ids, rand_ids=rn, sim_seed=np.array(sim_seed), step=np.array(step), teleport_x_offset=np.array(float(xo)))
    np.savez_compressed(os.path.join(HERE, name), **flat(inp), **flat(out), **names)
    moved = int(((e.root_states[:, :2] - inp["root_states"][:, :2]).abs().max(1).values > 1.0).sum())
    print(name, "teleported", moved, "pushed", len(push_ids), "re-randomised", len(rn))


def gen_ref_adaptation_module():
    """The reference's PRETRAINED adaptation module (runs/gait-conditioned-agility/pretrain-v0/train/025417.456545/checkpoints/
    adaptation_module_latest.jit: 2100 -> 256 -> 128 -> 2, ELU; trained on Isaac Gym rollouts of its own policy) as plain
    arrays (fp32) for tools/adaptation_probe.py (a sim-to-sim indicator: does it recover friction / restitution from THIS
    simulator's observation histories?).  The run's parameters.pkl gives the observation layout it expects."""
    import io
    import pickle
    run = os.path.join(REF, "runs", "gait-conditioned-agility", "pretrain-v0", "train", "025417.456545")
    m = torch.jit.load(os.path.join(run, "checkpoints", "adaptation_module_latest.jit"), map_location="cpu")
    sd = {k: v.detach().numpy() for k, v in m.state_dict().items()}

    # /root/reference is untrusted DATA: the pickle may only build plain containers, numpy arrays and torch tensors (an allow-list;
    # any other global is refused instead of imported), embedded tensor blobs are read with weights_only=True
    ALLOWED = {("builtins", n) for n in ("dict", "list", "tuple", "set", "frozenset", "int", "float", "bool", "str", "bytes", "complex", "slice")} | {
        ("collections", "OrderedDict"), ("numpy", "ndarray"), ("numpy", "dtype"), ("numpy.core.multiarray", "_reconstruct"),
        ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "scalar"),
        ("torch._utils", "_rebuild_tensor_v2"), ("torch", "FloatStorage"), ("torch", "LongStorage"), ("torch", "DoubleStorage"),
        ("torch", "IntStorage"), ("torch", "BoolStorage")}

    class U(pickle.Unpickler):
        def find_class(self, module, name):
            if module == "torch.storage" and name == "_load_from_bytes":
                return lambda b: torch.load(io.BytesIO(b), map_location="cpu", weights_only=True)
            if (module, name) not in ALLOWED:
                raise pickle.UnpicklingError(f"refusing to import {module}.{name} from the reference's parameters.pkl")
            return super().find_class(module, name)
    cfg = U(open(os.path.join(run, "parameters.pkl"), "rb")).load()["Cfg"]
    x = torch.randn(64, 2100, generator=torch.Generator().manual_seed(0))
    np.savez_compressed(os.path.join(HERE, "ref_adaptation_module.npz"),
                        **{"w" + k.replace(".", "_"): v.astype(np.float32) for k, v in sd.items()},
                        probe_in=x.numpy(), probe_out=m(x).detach().numpy(),
                        num_observations=np.array(cfg["env"]["num_observations"]), history=np.array(cfg["env"]["num_observation_history"]),
                        friction_range=np.array(cfg["domain_rand"]["friction_range"]), restitution_range=np.array(cfg["domain_rand"]["restitution_range"]),
                        norm_friction_range=np.array(cfg["normalization"]["friction_range"]),
                        norm_restitution_range=np.array(cfg["normalization"]["restitution_range"]))
    print("ref adaptation module", {k: v.shape for k, v in sd.items()})


if __name__ == "__main__":
    install_stubs()
    torch.manual_seed(0)
    if len(sys.argv) > 1 and sys.argv[1] == "callbacks":            # only callbacks*.npz
        gen_callbacks()
        gen_callbacks(seed=52, x_offset_px=163, name="callbacks_offset.npz")      # an evaluation region's teleport window
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "terrain_layout":       # only terrain_layout.npz
        gen_terrain_layout()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "pretrain_parameters":  # only pretrain_parameters.json
        gen_pretrain_parameters()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "traj_utils":           # only traj_utils.npz
        gen_traj_utils()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "sum_curriculum":       # only sum_curriculum.npz
        gen_sum_curriculum()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "recurrent_batches":    # only recurrent_batches.npz
        gen_recurrent_batches()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "cfg_defaults":         # only cfg_defaults.json
        gen_cfg_defaults()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "history_trace":        # only history_trace.json
        gen_history_trace()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "pretrain_jit_layout":  # only pretrain_jit_layout.json
        gen_pretrain_jit_layout()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "rollout":              # only rollout.npz
        gen_rollout()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "heights":              # only heights*.npz
        gen_heights()
        for m in [k for k in sys.modules if k.startswith("go1_gym")]:
            del sys.modules[m]
        gen_heights(seed=10, N=16, name="heights_coarse.npz", border=5.0, hscale=0.25, vscale=0.01, rows=70, cols=50,
                    points=([-0.6, -0.3, 0.0, 0.3, 0.6], [-0.25, 0.0, 0.25]))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "torques":              # only torques_<variant>.npz of the named variant
        gen_torques(sys.argv[2])
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ref_adaptation_module":
        gen_ref_adaptation_module()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ppo_fuzz":             # only ppo_fuzz<k>.npz
        gen_ppo_fuzz()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "maps_fuzz":            # only maps_fuzz<k>_mild.npz
        gen_maps_fuzz()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "maps_noise":           # only maps_train_noise.npz
        gen_maps("train_noise", seed=29, mild=True, noise=(777, 321))
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "maps":                 # only maps_<variant>[_mild].npz of the named variant
        gen_maps(sys.argv[2])
        for m in [k for k in sys.modules if k.startswith("go1_gym")]:
            del sys.modules[m]
        gen_maps(sys.argv[2], seed=23, mild=True)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "gravity":              # only gravity.npz
        gen_gravity()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "reset_eval":           # only reset_eval.npz
        gen_reset_eval()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "resample":            # only the resample_*.npz fixtures
        for mode in RESAMPLE_MODES:
            for m in [k for k in sys.modules if k.startswith("go1_gym")]:
                del sys.modules[m]
            gen_resample(mode)
        for m in [k for k in sys.modules if k.startswith("go1_gym")]:
            del sys.modules[m]
        gen_reset()
        for m in [k for k in sys.modules if k.startswith("go1_gym")]:
            del sys.modules[m]
        gen_reset_eval()
        sys.exit(0)
    gen_curriculum()
    gen_ppo()
    for m in [k for k in sys.modules if k.startswith("go1_gym")]:
        del sys.modules[m]
    gen_ppo_rma()
    gen_ppo_fuzz()
    for m in [k for k in sys.modules if k.startswith("go1_gym")]:
        del sys.modules[m]
    gen_rollout()
    for m in [k for k in sys.modules if k.startswith("go1_gym")]:
        del sys.modules[m]
    gen_heights()
    for m in [k for k in sys.modules if k.startswith("go1_gym")]:
        del sys.modules[m]
    gen_heights(seed=10, N=16, name="heights_coarse.npz", border=5.0, hscale=0.25, vscale=0.01, rows=70, cols=50,
                points=([-0.6, -0.3, 0.0, 0.3, 0.6], [-0.25, 0.0, 0.25]))
    for m in [k for k in sys.modules if k.startswith("go1_gym")]:
        del sys.modules[m]
    for v in ("train", "alt", "alt2"):
        # the reference mutates the global Cfg: one process state per variant
        for m in [k for k in sys.modules if k.startswith("go1_gym")]:
            del sys.modules[m]
        gen_maps(v)
        for m in [k for k in sys.modules if k.startswith("go1_gym")]:
            del sys.modules[m]
        gen_maps(v, seed=23, mild=True)
        for m in [k for k in sys.modules if k.startswith("go1_gym")]:
            del sys.modules[m]
        if v != "alt2":                                           # (alt2 changes observations only: its torque model is train's)
            gen_torques(v)
    for m in [k for k in sys.modules if k.startswith("go1_gym")]:
        del sys.modules[m]
    gen_maps("train_noise", seed=29, mild=True, noise=(777, 321))
    gen_maps_fuzz()
    for v in ("act_nolag", "pd_lag"):
        for m in [k for k in sys.modules if k.startswith("go1_gym")]:
            del sys.modules[m]
        gen_torques(v)
    for mode in RESAMPLE_MODES:
        for m in [k for k in sys.modules if k.startswith("go1_gym")]:
            del sys.modules[m]
        gen_resample(mode)
    for m in [k for k in sys.modules if k.startswith("go1_gym")]:
        del sys.modules[m]
    gen_reset()
    for m in [k for k in sys.modules if k.startswith("go1_gym")]:
        del sys.modules[m]
    gen_reset_eval()
    for m in [k for k in sys.modules if k.startswith("go1_gym")]:
        del sys.modules[m]
    gen_callbacks()
    gen_callbacks(seed=52, x_offset_px=163, name="callbacks_offset.npz")
    for m in [k for k in sys.modules if k.startswith("go1_gym")]:
        del sys.modules[m]
    gen_gravity()
    for m in [k for k in sys.modules if k.startswith("go1_gym")]:
        del sys.modules[m]
    gen_terrain_layout()
    gen_pretrain_parameters()
    gen_traj_utils()
    gen_sum_curriculum()
    gen_recurrent_batches()
    for m in [k for k in sys.modules if k.startswith("go1_gym")]:
        del sys.modules[m]
    gen_cfg_defaults()
    gen_history_trace()
    gen_pretrain_jit_layout()
    import subprocess
    subprocess.check_call([sys.executable, os.path.join(HERE, "gen_runner_iteration.py")])      # (its own interpreter: two go1_gym_learn packages)
    subprocess.check_call([sys.executable, os.path.join(HERE, "gen_runner_iteration.py"), "eval"])
