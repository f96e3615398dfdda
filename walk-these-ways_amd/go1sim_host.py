"""Host side of the Go1 step C-ABI: `Cfg` -> `Go1SimConfig`, buffer allocation, library loading.

This is the host mirror of what the reference computes once in `LeggedRobot.__init__`
(go1_gym/envs/base/legged_robot.py:41-53: `_parse_cfg` :1716-1732, `_init_buffers` :1123-1258,
`_prepare_reward_function` :1385-1429, `_init_command_distribution` :1299-1383,
`_get_noise_scale_vec` :1053-1120, `_process_dof_props` :581-609), flattened into the plain-C
config struct of include/go1sim.h.  PyTorch is used only to own device memory.
"""
import ctypes
import math
import os

import numpy as np
import torch

import go1sim_abi as abi
from go1_gym.envs.base.curriculum import RewardThresholdCurriculum

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libgo1sim.so")

LEGS = ["FL", "FR", "RL", "RR"]
BODY_NAMES = ["base"] + [f"{leg}_{part}" for leg in LEGS for part in ("hip", "thigh", "calf", "foot")]
DOF_NAMES = [f"{leg}_{part}_joint" for leg in LEGS for part in ("hip", "thigh", "calf")]

# URDF limits (resources/robots/go1/urdf/go1.urdf:96,138,166), order hip, thigh, calf
_DOF_LOWER = [-0.802851455917, -1.0471975512, -2.69653369433] * 4
_DOF_UPPER = [0.802851455917, 4.18879020479, -0.916297857297] * 4
_DOF_EFFORT = [33.5] * 12

COMMAND_KEYS = ["x_vel", "y_vel", "yaw_vel", "body_height", "gait_frequency", "gait_phase", "gait_offset",
                "gait_bounds", "gait_duration", "footswing_height", "body_pitch", "body_roll", "stance_width",
                "stance_length", "aux_reward_coef"]
_LIMIT_ATTR = ["limit_vel_x", "limit_vel_y", "limit_vel_yaw", "limit_body_height", "limit_gait_frequency",
               "limit_gait_phase", "limit_gait_offset", "limit_gait_bound", "limit_gait_duration",
               "limit_footswing_height", "limit_body_pitch", "limit_body_roll", "limit_stance_width",
               "limit_stance_length", "limit_aux_reward_coef"]
_BINS_ATTR = ["num_bins_vel_x", "num_bins_vel_y", "num_bins_vel_yaw", "num_bins_body_height",
              "num_bins_gait_frequency", "num_bins_gait_phase", "num_bins_gait_offset", "num_bins_gait_bound",
              "num_bins_gait_duration", "num_bins_footswing_height", "num_bins_body_pitch", "num_bins_body_roll",
              "num_bins_stance_width", "num_bins_stance_length", "num_bins_aux_reward_coef"]
_RANGE_ATTR = ["lin_vel_x", "lin_vel_y", "ang_vel_yaw", "body_height_cmd", "gait_frequency_cmd_range",
               "gait_phase_cmd_range", "gait_offset_cmd_range", "gait_bound_cmd_range", "gait_duration_cmd_range",
               "footswing_height_range", "body_pitch_range", "body_roll_range", "stance_width_range",
               "stance_length_range", "aux_reward_coef_range"]
LOCAL_RANGE = np.array([0.55, 0.55, 0.55, 0.55, 0.35, 0.25, 0.25, 0.25, 0.25, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0])
CURRICULUM_KEYS = ["tracking_lin_vel", "tracking_ang_vel", "tracking_contacts_shaped_force",
                   "tracking_contacts_shaped_vel"]
FAULT_NAMES = {v: k[len("GO1_FAULT_"):].lower() for k, v in abi.CONSTS.items()
               if k.startswith("GO1_FAULT_") and k != "GO1_FAULT_BITS"}
FAULT_FATAL_MASK = 0x3FF          # GO1_FAULT_FATAL_MASK of include/go1sim.h
CONTACT_CLASS_NAMES = {v: k[len("GO1_CC_"):].lower() for k, v in abi.CONSTS.items() if k.startswith("GO1_CC_") and k != "GO1_CC_COUNT"}
COMMAND_SUM_EXTRA = ["lin_vel_raw", "ang_vel_raw", "lin_vel_residual", "ang_vel_residual", "ep_timesteps"]


def policy_dt(cfg):
    """`self.dt` of the reference: decimation * SimParams.dt, the latter stored as float32."""
    return cfg.control.decimation * float(np.float32(cfg.sim.dt))


def _fill(arr, values):
    for i, v in enumerate(values):
        arr[i] = v


def build_curricula(cfg):
    """`_init_command_distribution` (legged_robot.py:1299-1383): one curriculum per gait category."""
    c = cfg.commands
    names = ['pronk', 'trot', 'pace', 'bound'] if c.gaitwise_curricula else ['nominal']
    ranges = {k: (getattr(c, la)[0], getattr(c, la)[1], getattr(c, ba))
              for k, la, ba in zip(COMMAND_KEYS, _LIMIT_ATTR, _BINS_ATTR)}
    low = np.array([getattr(c, a)[0] for a in _RANGE_ATTR], dtype=float)
    high = np.array([getattr(c, a)[1] for a in _RANGE_ATTR], dtype=float)
    curricula = []
    for _ in names:
        cur = RewardThresholdCurriculum(seed=c.curriculum_seed, **ranges)
        cur.set_to(low=low, high=high)
        curricula.append(cur)
    return names, curricula


def active_rewards(cfg):
    """`_prepare_reward_function` (legged_robot.py:1385-1413): non-zero scales in dict order, times dt."""
    dt = policy_dt(cfg)
    names, scales = [], []
    for name, scale in vars(cfg.reward_scales).items():
        if scale == 0:
            continue
        if name == "termination":
            raise NotImplementedError("reward 'termination' has no function in the reference's CoRLRewards")
        names.append(name)
        scales.append(scale * dt)
    return names, scales


def physx_section(cfg):
    """`cfg.sim.physx` as an attribute bag: a config class normally, a plain dict after scripts/play.py has restored the
    configuration from `parameters.pkl` (`setattr(Cfg.sim, "physx", {...})`, play.py:43-46)."""
    px = cfg.sim.physx
    if isinstance(px, dict):
        import types
        px = types.SimpleNamespace(**px)
    return px


def build_sim_config(cfg, num_envs=None, seed=0, env_id_offset=0, device_curriculum=True,
                     solver_iterations=4, warm_start=True, defer_curriculum_update=False, curriculum_update_interval=None):
    """Flatten `cfg` (a Cfg tree) into a Go1SimConfig.  Returns (struct, meta)."""
    S = abi.Go1SimConfig()
    S.abi_version = abi.GO1SIM_ABI_VERSION
    S.num_envs = int(num_envs if num_envs is not None else cfg.env.num_envs)
    S.seed = int(seed)
    S.env_id_offset = int(env_id_offset)
    dt = policy_dt(cfg)

    ctl = cfg.control
    S.decimation = ctl.decimation
    S.sim_dt = cfg.sim.dt
    if ctl.control_type not in ("P", "actuator_net"):
        raise NameError(f"Unknown controller type: {ctl.control_type}")   # legged_robot.py:943
    S.control_type = 1 if ctl.control_type == "actuator_net" else 0
    S.action_scale = ctl.action_scale
    S.hip_scale_reduction = ctl.hip_scale_reduction
    S.clip_actions = cfg.normalization.clip_actions
    kp = kd = 0.0
    for key in ctl.stiffness:                                   # legged_robot.py:1226-1230 (substring match)
        if key in "FL_hip_joint":
            kp, kd = ctl.stiffness[key], ctl.damping[key]
    S.kp, S.kd = kp, kd
    S.use_lag = int(cfg.domain_rand.randomize_lag_timesteps)
    S.lag_timesteps = int(cfg.domain_rand.lag_timesteps)
    assert S.lag_timesteps + 1 <= abi.GO1_MAX_LAG
    default = [cfg.init_state.default_joint_angles[n] for n in DOF_NAMES]
    _fill(S.default_dof_pos, default)
    _fill(S.torque_limits, _DOF_EFFORT)
    soft = cfg.rewards.soft_dof_pos_limit
    for i in range(12):                                         # legged_robot.py:603-607 (float32 arithmetic)
        lo, hi = np.float32(_DOF_LOWER[i]), np.float32(_DOF_UPPER[i])
        m, r = (lo + hi) / 2, hi - lo
        S.dof_pos_soft_lower[i] = float(m - np.float32(0.5) * r * np.float32(soft))
        S.dof_pos_soft_upper[i] = float(m + np.float32(0.5) * r * np.float32(soft))

    px = physx_section(cfg)
    _fill(S.gravity, [0.0, 0.0, -9.8])                          # legged_robot.py:558
    S.contact_distance = 2.0 * px.contact_offset
    S.max_depenetration_velocity = px.max_depenetration_velocity
    S.bounce_threshold_velocity = px.bounce_threshold_velocity
    S.terrain_friction = cfg.terrain.static_friction
    S.terrain_dynamic_friction = getattr(cfg.terrain, "dynamic_friction", cfg.terrain.static_friction)
    S.hf_wall_units = 0                                         # set by bind_height_field(slope_threshold=...) for a trimesh terrain
    S.terrain_restitution = cfg.terrain.restitution
    S.max_linear_velocity = float(cfg.asset.max_linear_velocity)
    S.max_angular_velocity = float(cfg.asset.max_angular_velocity)
    S.joint_limit_margin = float(getattr(px, "joint_limit_margin", 5.0))
    S.joint_limit_pos_margin = float(getattr(px, "joint_limit_pos_margin", 0.1))
    S.self_collision = int(getattr(cfg.asset, "self_collisions", 0) == 0)       # the flag is a collision FILTER: 0 = enabled
    S.solver_iterations = int(solver_iterations)
    S.warm_start = int(warm_start)
    S.terrain_type = 0                                          # set by bind_height_field() when a height field is bound
    S.measure_heights = int(ter_measure(cfg))
    xs, ys = list(cfg.terrain.measured_points_x), list(cfg.terrain.measured_points_y)
    assert len(xs) <= abi.GO1_MAX_HEIGHT_AXIS and len(ys) <= abi.GO1_MAX_HEIGHT_AXIS
    S.num_height_x, S.num_height_y = len(xs), len(ys)
    _fill(S.height_points_x, xs)
    _fill(S.height_points_y, ys)
    S.observe_heights = int(getattr(cfg.env, "observe_heights", False) and S.measure_heights)
    S.obs_scale_height = cfg.obs_scales.height_measurements
    S.height_noise_scale = cfg.noise_scales.height_measurements * cfg.noise.noise_level * cfg.obs_scales.height_measurements

    S.max_episode_length = int(math.ceil(cfg.env.episode_length_s / dt))
    S.resample_interval = int(cfg.commands.resampling_time / dt)
    S.rand_interval = int(math.ceil(cfg.domain_rand.rand_interval_s / dt))
    dr = cfg.domain_rand
    S.randomize_gravity = int(dr.randomize_gravity)
    S.gravity_rand_interval = int(math.ceil(dr.gravity_rand_interval_s / dt))
    S.gravity_rand_duration = int(math.ceil(S.gravity_rand_interval * dr.gravity_impulse_duration))
    _fill(S.gravity_range, dr.gravity_range)
    S.push_robots = int(dr.push_robots)
    S.push_interval = int(math.ceil(dr.push_interval_s / dt))
    S.max_push_vel_xy = dr.max_push_vel_xy
    S.randomize_motor_strength = int(dr.randomize_motor_strength)
    S.randomize_motor_offset = int(getattr(dr, "randomize_motor_offset", False))
    S.randomize_Kp_factor = int(dr.randomize_Kp_factor)
    S.randomize_Kd_factor = int(dr.randomize_Kd_factor)
    _fill(S.motor_strength_range, dr.motor_strength_range)
    _fill(S.motor_offset_range, getattr(dr, "motor_offset_range", [0.0, 0.0]))
    _fill(S.Kp_factor_range, dr.Kp_factor_range)
    _fill(S.Kd_factor_range, dr.Kd_factor_range)
    S.randomize_rigids_after_start = int(getattr(dr, "randomize_rigids_after_start", False))
    S.randomize_base_mass, S.randomize_com_displacement = int(dr.randomize_base_mass), int(dr.randomize_com_displacement)
    S.randomize_friction, S.randomize_restitution = int(dr.randomize_friction), int(dr.randomize_restitution)
    _fill(S.added_mass_range, dr.added_mass_range)
    _fill(S.com_displacement_range, dr.com_displacement_range)
    _fill(S.friction_range, dr.friction_range)
    _fill(S.restitution_range, dr.restitution_range)
    ter = cfg.terrain
    custom_origins = ter.mesh_type in ("heightfield", "trimesh")
    S.teleport_robots = int(ter.teleport_robots and custom_origins)
    S.teleport_thresh = ter.teleport_thresh
    S.teleport_x_offset = float(int(getattr(ter, "x_offset", 0) * ter.horizontal_scale))
    S.terrain_length, S.terrain_width = ter.terrain_length, ter.terrain_width
    S.terrain_num_rows, S.terrain_num_cols = ter.num_rows, ter.num_cols

    ist = cfg.init_state
    _fill(S.base_init_state, list(ist.pos) + list(ist.rot) + list(ist.lin_vel) + list(ist.ang_vel))
    S.custom_origins = int(custom_origins)
    S.x_init_range, S.y_init_range, S.yaw_init_range = ter.x_init_range, ter.y_init_range, ter.yaw_init_range
    S.x_init_offset, S.y_init_offset = ter.x_init_offset, ter.y_init_offset

    def mask(patterns):
        m = 0
        for p in patterns:
            for i, n in enumerate(BODY_NAMES):
                if p in n:
                    m |= 1 << i
        return m
    S.termination_body_mask = mask(cfg.asset.terminate_after_contacts_on)
    S.penalised_body_mask = mask(cfg.asset.penalize_contacts_on)
    S.use_terminal_body_height = int(cfg.rewards.use_terminal_body_height)
    S.terminal_body_height = cfg.rewards.terminal_body_height

    env = cfg.env
    S.num_obs, S.num_privileged_obs, S.num_obs_history = env.num_observations, env.num_privileged_obs, env.num_observation_history
    S.num_commands = cfg.commands.num_commands
    for flag in ("observe_command", "observe_two_prev_actions", "observe_timing_parameter", "observe_clock_inputs",
                 "observe_vel", "observe_only_ang_vel", "observe_only_lin_vel", "observe_yaw", "observe_contact_states",
                 "observe_gait_commands"):
        setattr(S, flag, int(getattr(env, flag)))
    S.global_reference = int(cfg.commands.global_reference)
    S.pacing_offset = int(cfg.commands.pacing_offset)
    os_ = cfg.obs_scales
    S.obs_scale_lin_vel, S.obs_scale_ang_vel = os_.lin_vel, os_.ang_vel
    S.obs_scale_dof_pos, S.obs_scale_dof_vel = os_.dof_pos, os_.dof_vel
    cs = [os_.lin_vel, os_.lin_vel, os_.ang_vel, os_.body_height_cmd, os_.gait_freq_cmd, os_.gait_phase_cmd,
          os_.gait_phase_cmd, os_.gait_phase_cmd, os_.gait_phase_cmd, os_.footswing_height_cmd, os_.body_pitch_cmd,
          os_.body_roll_cmd, os_.stance_width_cmd, os_.stance_length_cmd, os_.aux_reward_cmd]   # legged_robot.py:1196-1203
    _fill(S.commands_scale, cs)
    S.add_noise = int(cfg.noise.add_noise)
    nv = noise_scale_vec(cfg)
    scan = S.num_height_x * S.num_height_y if S.observe_heights else 0
    if len(nv) > abi.GO1_MAX_OBS:
        raise ValueError("scalar observations exceed GO1_MAX_OBS")
    if len(nv) + scan != env.num_observations:
        raise ValueError(f"num_observations ({env.num_observations}) != assembled observation width ({len(nv) + scan})")
    _fill(S.noise_scale_vec, nv)
    S.clip_observations = cfg.normalization.clip_observations
    nz = cfg.normalization
    priv = {"friction": ("priv_observe_friction", nz.friction_range),
            "restitution": ("priv_observe_restitution", nz.restitution_range),
            "base_mass": ("priv_observe_base_mass", nz.added_mass_range),
            "com_displacement": ("priv_observe_com_displacement", nz.com_displacement_range),
            "motor_strength": ("priv_observe_motor_strength", nz.motor_strength_range),
            "motor_offset": ("priv_observe_motor_offset", nz.motor_offset_range),
            "body_height": ("priv_observe_body_height", nz.body_height_range),
            "body_velocity": ("priv_observe_body_velocity", nz.body_velocity_range),
            "gravity": ("priv_observe_gravity", nz.gravity_range),
            "clock_inputs": ("priv_observe_clock_inputs", [-1, 1]),
            "desired_contact": ("priv_observe_desired_contact_states", [-1, 1])}
    width = {"com_displacement": 3, "motor_strength": 12, "motor_offset": 12, "body_velocity": 3, "gravity": 3,
             "clock_inputs": 4, "desired_contact": 4}
    npriv = 0
    for name, (flag, rng) in priv.items():
        idx = abi.PRIV_IDS[name]
        on = bool(getattr(env, flag, False))
        S.priv_enabled[idx] = int(on)
        S.priv_scale[idx] = 2.0 / (rng[1] - rng[0])              # math_utils.py:35-38
        S.priv_shift[idx] = (rng[1] + rng[0]) / 2.0
        npriv += width.get(name, 1) if on else 0
    if getattr(env, "priv_observe_ground_friction", False):
        raise NotImplementedError("priv_observe_ground_friction calls an undefined method in the reference (App. D8)")
    assert npriv == env.num_privileged_obs, (
        f"num_privileged_obs ({env.num_privileged_obs}) != the number of privileged observations ({npriv}), "
        f"you will discard data from the student!")                # legged_robot.py:490-491
    assert npriv <= abi.GO1_MAX_PRIV_OBS

    names, scales = active_rewards(cfg)
    assert len(names) <= abi.GO1_MAX_REWARDS
    S.num_rewards = len(names)
    for i, (n, s) in enumerate(zip(names, scales)):
        if n not in abi.REWARD_IDS:
            print(f"Warning: reward {'_reward_' + n} has nonzero coefficient but was not found!")
        S.reward_ids[i] = abi.REWARD_IDS.get(n, -1)
        S.reward_scales[i] = s
    rw = cfg.rewards
    S.only_positive_rewards = int(rw.only_positive_rewards)
    S.only_positive_rewards_ji22_style = int(rw.only_positive_rewards_ji22_style)
    S.sigma_rew_neg = rw.sigma_rew_neg
    S.dt = dt
    S.tracking_sigma, S.tracking_sigma_yaw = rw.tracking_sigma, rw.tracking_sigma_yaw
    S.base_height_target, S.max_contact_force = rw.base_height_target, rw.max_contact_force
    S.kappa_gait_probs, S.gait_force_sigma, S.gait_vel_sigma = rw.kappa_gait_probs, rw.gait_force_sigma, rw.gait_vel_sigma
    # NOT a reference field (absent = the reference's world-z reward terms): include/go1sim.h reward_heights_above_terrain
    S.reward_heights_above_terrain = int(bool(getattr(rw, "heights_above_terrain", False)))

    cm = cfg.commands
    S.device_curriculum = int(device_curriculum)
    S.defer_curriculum_update = int(defer_curriculum_update)
    K = curriculum_update_interval if curriculum_update_interval is not None else getattr(cm, "curriculum_update_interval", 1)
    S.curriculum_update_interval = int(K)
    assert 1 <= S.curriculum_update_interval <= abi.GO1_MAX_CURRICULUM_INTERVAL
    S.gaitwise_curricula = int(cm.gaitwise_curricula)
    S.binary_phases = int(cm.binary_phases)
    S.exclusive_phase_offset = int(getattr(cm, "exclusive_phase_offset", False))
    S.balance_gait_distribution = int(getattr(cm, "balance_gait_distribution", False))
    cat_names, curricula = build_curricula(cfg)
    S.num_categories = len(cat_names)
    S.num_bins = len(curricula[0])
    for i, (la, ba) in enumerate(zip(_LIMIT_ATTR, _BINS_ATTR)):
        S.grid_bins[i] = getattr(cm, ba)
        S.grid_low[i], S.grid_high[i] = getattr(cm, la)
    thr = vars(cfg.curriculum_thresholds)
    keys = 0
    for k, key in enumerate(CURRICULUM_KEYS):
        if key in names:                                        # legged_robot.py:728-732
            keys |= 1 << k
            S.curriculum_sum_index[k] = names.index(key)
            S.curriculum_threshold[k] = thr[key] * scales[names.index(key)]
    S.curriculum_keys = keys

    meta = dict(reward_names=names, reward_scales=dict(zip(names, scales)), category_names=cat_names,
                curricula=curricula, dt=dt,
                episode_sum_names=names + ["total"], command_sum_names=names + COMMAND_SUM_EXTRA)
    return S, meta


def ter_measure(cfg):
    return bool(cfg.terrain.measure_heights)


def bind_height_field(S, buffers, heights_int16, hscale, vscale, border, slope_threshold=None):
    """`_create_heightfield` / `_create_trimesh` (legged_robot.py:1441-1479): hand the int16 height samples to the
    simulator.  A constant field is served by the plane fast path (same physics, no gathers).  `slope_threshold` (the
    trimesh terrain's `slope_treshold`, terrain.py:33-36): cell edges rising by more than slope_threshold * hscale become
    vertical faces (include/go1sim.h hf_wall_units) — only when the field has such an edge at all."""
    hs = torch.as_tensor(np.ascontiguousarray(heights_int16), dtype=torch.int16)
    S.hf_rows, S.hf_cols = int(hs.shape[0]), int(hs.shape[1])
    S.hf_hscale, S.hf_vscale, S.hf_border = float(hscale), float(vscale), float(border)
    flat = bool((hs == hs.flatten()[0]).all()) and int(hs.flatten()[0]) == 0
    S.terrain_type = 0 if flat else 1
    S.hf_wall_units = 0
    if slope_threshold is not None and not flat:
        # the reference compares int16 sample differences with slope_threshold * (horizontal_scale / vertical_scale), a Python
        # float (isaacgym.terrain_utils.convert_heightfield_to_trimesh): the largest difference that is NOT steep
        units = int(np.floor(float(slope_threshold) * (float(hscale) / float(vscale))))
        h = hs.to(torch.int32)
        steepest = max(int((h[1:] - h[:-1]).abs().max()), int((h[:, 1:] - h[:, :-1]).abs().max()))
        if steepest > units:
            S.hf_wall_units = max(units, 1)
    buffers.tensors["height_samples"] = hs.to(buffers.device)
    buffers.refresh_struct()
    return S


def noise_scale_vec(cfg):
    """`_get_noise_scale_vec` (legged_robot.py:1053-1120)."""
    env, ns, lvl, os_ = cfg.env, cfg.noise_scales, cfg.noise.noise_level, cfg.obs_scales
    na = env.num_actions
    vec = [ns.gravity * lvl] * 3
    if env.observe_command:
        vec += [0.0] * cfg.commands.num_commands
    vec += [ns.dof_pos * lvl * os_.dof_pos] * na + [ns.dof_vel * lvl * os_.dof_vel] * na + [0.0] * na
    if env.observe_two_prev_actions:
        vec += [0.0] * na
    if env.observe_timing_parameter:
        vec += [0.0]
    if env.observe_clock_inputs:
        vec += [0.0] * 4
    if env.observe_vel:
        vec = [ns.lin_vel * lvl * os_.lin_vel] * 3 + [ns.ang_vel * lvl * os_.ang_vel] * 3 + vec
    if env.observe_only_lin_vel:
        vec = [ns.lin_vel * lvl * os_.lin_vel] * 3 + vec
    if env.observe_yaw:
        vec += [0.0]
    if env.observe_contact_states:
        vec += [ns.contact_states * lvl] * 4
    return vec


# ---------------------------------------------------------------------------------------------
def _buffer_specs(S):
    N, nr, nl = S.num_envs, S.num_rewards, S.lag_timesteps + 1
    f, i32, u8 = torch.float32, torch.int32, torch.uint8
    nb, nc, ki = S.num_bins, S.num_categories, max(int(S.curriculum_update_interval), 1)
    return {
        "root_states": (f, (13, N)), "dof_pos": (f, (12, N)), "dof_vel": (f, (12, N)),
        "contact_forces": (f, (51, N)), "foot_positions": (f, (12, N)), "foot_velocities": (f, (12, N)),
        "prev_foot_velocities": (f, (12, N)),
        "lag_buffer": (f, (nl, 12, N)),
        "joint_pos_err_last": (f, (12, N)), "joint_pos_err_last_last": (f, (12, N)),
        "joint_vel_last": (f, (12, N)), "joint_vel_last_last": (f, (12, N)),
        "joint_pos_target": (f, (12, N)), "last_joint_pos_target": (f, (12, N)),
        "last_last_joint_pos_target": (f, (12, N)),
        "actions": (f, (12, N)), "last_actions": (f, (12, N)), "last_last_actions": (f, (12, N)),
        "last_dof_vel": (f, (12, N)), "torques": (f, (12, N)),
        "base_lin_vel": (f, (3, N)), "base_ang_vel": (f, (3, N)), "projected_gravity": (f, (3, N)),
        "commands": (f, (abi.GO1_MAX_COMMANDS, N)), "gait_indices": (f, (N,)),
        "clock_inputs": (f, (4, N)), "desired_contact_states": (f, (4, N)), "foot_indices": (f, (4, N)),
        "episode_length_buf": (i32, (N,)), "reset_buf": (u8, (N,)), "time_out_buf": (u8, (N,)),
        "last_contacts": (u8, (4, N)), "resample_flags": (u8, (N,)),
        "rew_buf": (f, (N,)), "episode_sums": (f, (nr + 1, N)), "command_sums": (f, (nr + 5, N)),
        "episode_log": (f, (nr + 2,)), "episode_sums_eval": (f, (nr + 1, N)),
        "friction_coeffs": (f, (N,)), "restitutions": (f, (N,)), "payloads": (f, (N,)),
        "com_displacements": (f, (3, N)), "motor_strengths": (f, (12, N)), "motor_offsets": (f, (12, N)),
        "Kp_factors": (f, (12, N)), "Kd_factors": (f, (12, N)), "env_origins": (f, (3, N)),
        "env_command_bins": (i32, (N,)), "env_command_categories": (i32, (N,)),
        "curriculum_weights": (f, (nc, nb)), "curriculum_cdf": (f, (nc, nb)), "curriculum_success": (i32, (ki, nc, nb)),
        "obs_buf": (f, (N, S.num_obs)), "privileged_obs_buf": (f, (N, max(S.num_privileged_obs, 1))),
        "obs_history": (f, (N, 2 * (S.num_obs_history + 1) * S.num_obs)),
        "measured_heights": (f, (max(S.num_height_x * S.num_height_y, 1), N)),
        "fault_flags": (i32, (N,)), "fault_counts": (i32, (abi.GO1_FAULT_BITS,)),
        "contact_drop_counts": (i32, (abi.GO1_CC_COUNT,)),
    }


class SimBuffers:
    """Owns one torch tensor per Go1SimBuffers field (SoA layouts of include/go1sim.h) on `device`."""

    def __init__(self, S, meta, device="cpu"):
        self.device = torch.device(device)
        self.tensors = {}
        for name, (dtype, shape) in _buffer_specs(S).items():
            self.tensors[name] = torch.zeros(shape, dtype=dtype, device=self.device)
        t = self.tensors
        t["root_states"][6].fill_(1.0)
        t["motor_strengths"].fill_(1.0)
        t["Kp_factors"].fill_(1.0)
        t["Kd_factors"].fill_(1.0)
        t["friction_coeffs"].fill_(1.0)
        cur = meta["curricula"]
        w = np.stack([c.weights for c in cur]).astype(np.float32)
        t["curriculum_weights"].copy_(torch.from_numpy(w))
        cdf = np.cumsum(w.astype(np.float64), axis=1)
        cdf /= cdf[:, -1:]
        t["curriculum_cdf"].copy_(torch.from_numpy(cdf.astype(np.float32)))
        ptr, idx = cur[0].neighbourhood_csr(LOCAL_RANGE)
        t["curriculum_nbr_ptr"] = torch.from_numpy(ptr).to(self.device)
        t["curriculum_nbr_idx"] = torch.from_numpy(idx).to(self.device)
        t["height_samples"] = None
        t["contact_signature"] = None       # tests: enable_contact_signature()
        self.struct = abi.Go1SimBuffers()
        self.refresh_struct()

    def enable_contact_signature(self):
        """allocate the per-substep record of listed contact points / self pairs / limit-row legs (include/go1sim.h)"""
        n = self.tensors["root_states"].shape[1]
        self.tensors["contact_signature"] = torch.zeros(abi.GO1_SIG_MAX_SUBSTEPS * abi.GO1_SIG_WORDS, n, dtype=torch.int32, device=self.device)
        self.refresh_struct()
        return self.tensors["contact_signature"]

    def refresh_struct(self):
        for name in abi.BUFFER_FIELDS:
            ten = self.tensors.get(name)
            setattr(self.struct, name, None if ten is None else ten.data_ptr())

    def __getattr__(self, name):
        try:
            return self.__dict__["tensors"][name]
        except KeyError:
            raise AttributeError(name)

    def clone_to(self, device):
        other = object.__new__(SimBuffers)
        other.device = torch.device(device)
        other.tensors = {k: (None if v is None else v.detach().to(device).clone()) for k, v in self.tensors.items()}
        other.struct = abi.Go1SimBuffers()
        other.refresh_struct()
        return other


# The fields the reference reads from the environment's OWN group configuration (`_call_train_eval(func, env_ids)` ->
# func(env_ids, cfg | eval_cfg), legged_robot.py:531-544): _randomize_dof_props :646-665, _randomize_rigid_body_props
# :611-633, _push_robots :1020-1026, _teleport_robots :1031-1045, _reset_root_states :973-980.  Everything else
# (commands / curriculum, rewards, observations, control, physics) comes from the train configuration for all environments.
EVAL_CFG_FIELDS = ("randomize_motor_strength", "randomize_motor_offset", "randomize_Kp_factor", "randomize_Kd_factor",
                   "motor_strength_range", "motor_offset_range", "Kp_factor_range", "Kd_factor_range",
                   "randomize_base_mass", "randomize_com_displacement", "randomize_friction", "randomize_restitution",
                   "added_mass_range", "com_displacement_range", "friction_range", "restitution_range",
                   "push_robots", "push_interval", "max_push_vel_xy",
                   "teleport_robots", "teleport_thresh", "teleport_x_offset", "terrain_length", "terrain_width",
                   "terrain_num_rows", "terrain_num_cols",
                   "x_init_range", "y_init_range", "yaw_init_range", "x_init_offset", "y_init_offset")


def _philox4x32_10(ctr, key):
    """Philox-4x32-10 (Random123) — the generator of the kernels (csrc/go1_math.h), for the few values the host derives itself."""
    M0, M1, W0, W1, mask = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85, 0xFFFFFFFF
    c, k = [int(x) & mask for x in ctr], [int(x) & mask for x in key]
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & mask, (p0 >> 32) ^ c[3] ^ k[1], p0 & mask]
        k = [(k[0] + W0) & mask, (k[1] + W1) & mask]
    return c


def gravity_at(S, t):
    """Gravity vector in force once `t` policy steps have been taken (= during step t + 1): the nominal vector plus, for
    `gravity_rand_duration` steps out of every `gravity_rand_interval`, one offset drawn for ALL environments from
    U(gravity_range)^3 (reference `_randomize_gravity` legged_robot.py:546-561 on the cadence of :701-705; the kernels evaluate
    the same function of (seed, epoch), csrc/go1_maps.h `gravity_at`, so nothing has to be read back)."""
    g = np.array([S.gravity[0], S.gravity[1], S.gravity[2]], dtype=np.float32)
    if not S.randomize_gravity:
        return g
    epoch, phase = divmod(int(t), int(S.gravity_rand_interval))
    if phase >= int(S.gravity_rand_duration):
        return g
    seed = int(S.seed) & 0xFFFFFFFFFFFFFFFF
    word = _philox4x32_10((0xFFFFFFFF, epoch, 8, 0), (seed & 0xFFFFFFFF, seed >> 32))      # purpose 8 = gravity, columns 0..2
    lo, hi = np.float32(S.gravity_range[0]), np.float32(S.gravity_range[1])
    u = np.array([np.float32(w >> 8) * np.float32(1.0 / 16777216.0) for w in word[:3]], dtype=np.float32)
    return g + (u * (hi - lo) + lo)


def make_eval_sim_config(S_train, S_from_eval_cfg):
    """Go1SimConfig of the evaluation environments: the train block with the group-dispatched fields of `eval_cfg`
    (S_from_eval_cfg = build_sim_config(eval_cfg, ...))."""
    S = abi.Go1SimConfig()
    ctypes.memmove(ctypes.byref(S), ctypes.byref(S_train), ctypes.sizeof(S))
    for name in EVAL_CFG_FIELDS:
        v = getattr(S_from_eval_cfg, name)
        if hasattr(v, "__len__"):
            dst = getattr(S, name)
            for i in range(len(v)):
                dst[i] = v[i]
        else:
            setattr(S, name, v)
    return S


# ---------------------------------------------------------------------------------------------
_lib = None


class Go1SimLibraryMissing(RuntimeError):
    pass


def load_library():
    """Load libgo1sim.so (HIP, gfx950).  Fails loudly: there is no CPU fallback on the product path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Go1SimLibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(hipcc --offload-arch=gfx950). The Go1 step has no CPU fallback.")
    _lib = bind_library(ctypes.CDLL(LIB_PATH))
    return _lib


def bind_library(lib):
    """ctypes prototypes of include/go1sim.h on a loaded library object."""
    vp, i32, i64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64
    lib.go1sim_create.argtypes = [ctypes.POINTER(abi.Go1SimConfig), ctypes.POINTER(abi.Go1SimBuffers), ctypes.c_int, ctypes.POINTER(vp)]
    lib.go1sim_destroy.argtypes = [vp]
    lib.go1sim_set_config.argtypes = [vp, ctypes.POINTER(abi.Go1SimConfig)]
    lib.go1sim_set_eval_config.argtypes = [vp, ctypes.POINTER(abi.Go1SimConfig), i32]
    lib.go1sim_step.argtypes = [vp, vp, vp]
    lib.go1sim_reset_idx.argtypes = [vp, vp, i32, vp]
    lib.go1sim_compute_torques.argtypes = [vp, vp, vp]
    lib.go1sim_physics_substep.argtypes = [vp, vp]
    lib.go1sim_curriculum_update.argtypes = [vp, vp]
    lib.go1sim_post_physics.argtypes = [vp, vp, vp]
    lib.go1sim_append_history.argtypes = [vp, vp]
    lib.go1sim_history_window_offset.argtypes = [vp, ctypes.POINTER(i32)]
    lib.go1sim_get_counters.argtypes = [vp, ctypes.POINTER(i64), ctypes.POINTER(i32)]
    lib.go1sim_set_counters.argtypes = [vp, i64, i32]
    lib.go1sim_enable_timing.argtypes = [vp, ctypes.c_int]
    lib.go1sim_read_timings.argtypes = [vp, ctypes.POINTER(ctypes.c_float), i32, ctypes.POINTER(i32)]
    lib.go1sim_version.restype = ctypes.c_char_p
    for fn in ("go1sim_create", "go1sim_destroy", "go1sim_set_config", "go1sim_set_eval_config", "go1sim_step", "go1sim_reset_idx",
               "go1sim_compute_torques", "go1sim_physics_substep", "go1sim_curriculum_update", "go1sim_post_physics",
               "go1sim_append_history", "go1sim_history_window_offset",
               "go1sim_get_counters", "go1sim_set_counters", "go1sim_enable_timing", "go1sim_read_timings"):
        getattr(lib, fn).restype = ctypes.c_int
    return lib


EXPORTED_SYMBOLS = ["go1sim_create", "go1sim_destroy", "go1sim_set_config", "go1sim_set_eval_config", "go1sim_step", "go1sim_reset_idx",
                    "go1sim_compute_torques", "go1sim_physics_substep", "go1sim_curriculum_update",
                    "go1sim_post_physics", "go1sim_append_history", "go1sim_history_window_offset",
                    "go1sim_get_counters", "go1sim_set_counters", "go1sim_enable_timing",
                    "go1sim_read_timings", "go1sim_version"]


class Go1Sim:
    """Thin RAII wrapper over the opaque handle."""

    def __init__(self, S, buffers, device_index=0, lib=None):
        self.lib = lib if lib is not None else load_library()
        self.S, self.buffers = S, buffers
        self.handle = ctypes.c_void_p()
        rc = self.lib.go1sim_create(ctypes.byref(S), ctypes.byref(buffers.struct), int(device_index), ctypes.byref(self.handle))
        if rc != 0:
            raise RuntimeError(f"go1sim_create failed: {rc}")

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed: {rc}")

    def step(self, actions):
        assert actions.is_cuda and actions.dtype == torch.float32 and actions.is_contiguous()
        self._check(self.lib.go1sim_step(self.handle, ctypes.c_void_p(actions.data_ptr()), self._stream()), "go1sim_step")

    def reset_idx(self, ids=None):
        if ids is None:
            rc = self.lib.go1sim_reset_idx(self.handle, None, 0, self._stream())
        else:
            ids = ids.to(dtype=torch.int32, device=self.buffers.device).contiguous()
            self._keep = ids
            rc = self.lib.go1sim_reset_idx(self.handle, ctypes.c_void_p(ids.data_ptr()), ids.numel(), self._stream())
        self._check(rc, "go1sim_reset_idx")

    def compute_torques(self, actions_soa):
        self._check(self.lib.go1sim_compute_torques(self.handle, ctypes.c_void_p(actions_soa.data_ptr()), self._stream()), "go1sim_compute_torques")

    def physics_substep(self):
        self._check(self.lib.go1sim_physics_substep(self.handle, self._stream()), "go1sim_physics_substep")

    def curriculum_update(self):
        self._check(self.lib.go1sim_curriculum_update(self.handle, self._stream()), "go1sim_curriculum_update")

    def post_physics(self, gravity):
        g = (ctypes.c_float * 3)(*[float(x) for x in gravity])
        self._check(self.lib.go1sim_post_physics(self.handle, ctypes.cast(g, ctypes.c_void_p), self._stream()), "go1sim_post_physics")

    def append_history(self):
        self._check(self.lib.go1sim_append_history(self.handle, self._stream()), "go1sim_append_history")

    def history_window_offset(self):
        off = ctypes.c_int32()
        self.lib.go1sim_history_window_offset(self.handle, ctypes.byref(off))
        return off.value

    def set_config(self, S):
        """Replace the (train) configuration.  With an evaluation split installed the evaluation block is rebuilt as the new
        configuration plus the group-dispatched fields it carried (EVAL_CFG_FIELDS): every field the reference reads from the
        train configuration for ALL environments (rewards, commands, control, physics) follows the change."""
        self.S = S
        self._check(self.lib.go1sim_set_config(self.handle, ctypes.byref(S)), "go1sim_set_config")
        if getattr(self, "S_eval", None) is not None:
            self.set_eval_config(make_eval_sim_config(S, self.S_eval), self.num_train_envs)

    def set_eval_config(self, S_eval, num_train_envs):
        """environments [num_train_envs, N) run under S_eval (make_eval_sim_config); num_train_envs % 16 == 0"""
        self.S_eval, self.num_train_envs = S_eval, int(num_train_envs)
        self._check(self.lib.go1sim_set_eval_config(self.handle, ctypes.byref(S_eval), int(num_train_envs)), "go1sim_set_eval_config")

    def counters(self):
        c, h = ctypes.c_int64(), ctypes.c_int32()
        self.lib.go1sim_get_counters(self.handle, ctypes.byref(c), ctypes.byref(h))
        return c.value, h.value

    def set_counters(self, counter, lag_head):
        self.lib.go1sim_set_counters(self.handle, int(counter), int(lag_head))

    def enable_timing(self, capacity):
        self._check(self.lib.go1sim_enable_timing(self.handle, int(capacity)), "go1sim_enable_timing")

    def read_timings(self, max_n=65536):
        buf = (ctypes.c_float * max_n)()
        n = ctypes.c_int32()
        self._check(self.lib.go1sim_read_timings(self.handle, buf, max_n, ctypes.byref(n)), "go1sim_read_timings")
        return list(buf[:n.value])

    def __del__(self):
        try:
            if self.handle:
                self.lib.go1sim_destroy(self.handle)
                self.handle = None
        except Exception:
            pass
