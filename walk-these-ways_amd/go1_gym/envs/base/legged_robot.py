"""`LeggedRobot`: the vectorised Go1 environment on MI355X.

Same step / reset / get_observations surface, constructor and attribute names as the reference's
go1_gym/envs/base/legged_robot.py (`LeggedRobot` :19-58, `step` :60-88, `reset_idx` :150-239, buffers
:1123-1297), but nothing here computes per-environment arithmetic in Python: `step()` is one call into
libgo1sim (HIP, include/go1sim.h `go1sim_step`), which runs the torque model, the 4 physics substeps and
every tensor map of `post_physics_step` in a single kernel launch on the current torch stream, with no
host synchronisation.  This class owns the device buffers (torch tensors, SoA layout) and exposes them
under the reference's attribute names as (N, ...) views.
"""
import os
from collections.abc import Mapping

import numpy as np
import torch

import go1sim_host as H
from go1_gym.envs.base.base_task import BaseTask
from go1_gym.utils.terrain import Terrain


class _LazyMapping(Mapping):
    """Base of the `extras` entries that are evaluated on access.  Pickled (the reference Runner dumps
    `extras["curriculum/distribution"]` with `logger.save_pkl`, ppo_cse/__init__.py:200-208) and deep-copied as the plain dict
    of their current values — never as a reference to the environment."""

    def __reduce__(self):
        return (dict, (dict(self.items()),))


class _EpisodeStats(_LazyMapping):
    """`extras["train/episode"]` (reference legged_robot.py:181-227): means of the per-term episode sums over the
    environments that were reset, plus command-range statistics.  Evaluated lazily on access (device reductions,
    no work and no sync inside step()); the running sums are kept by the kernel in `episode_log`."""

    def __init__(self, env):
        self.env = env

    def _keys(self):
        e = self.env
        keys = ["rew_" + n for n in e.episode_sum_names]
        if e.cfg.commands.command_curriculum:
            for n in ("duration", "bound", "offset", "phase", "freq", "x_vel", "y_vel", "yaw_vel"):
                keys += [f"min_command_{n}", f"max_command_{n}"]
            if e.cfg.commands.num_commands > 9:
                keys += ["min_command_swing_height", "max_command_swing_height"]
            keys += [f"command_area_{c}" for c in e.category_names] + ["min_action", "max_action"]
        return keys

    def __iter__(self):
        return iter(self._keys())

    def __len__(self):
        return len(self._keys())

    def __getitem__(self, key):
        e = self.env
        log = e.buffers.episode_log
        if key.startswith("rew_"):
            i = e.episode_sum_names.index(key[4:])
            return log[i] / log[-1].clamp(min=1.0)
        col = {"duration": 8, "bound": 7, "offset": 6, "phase": 5, "freq": 4, "x_vel": 0, "y_vel": 1, "yaw_vel": 2,
               "swing_height": 9}
        if key.startswith("min_command_"):
            return e.buffers.commands[col[key[12:]]].min()
        if key.startswith("max_command_"):
            return e.buffers.commands[col[key[12:]]].max()
        if key.startswith("command_area_"):
            w = e.buffers.curriculum_weights[e.category_names.index(key[13:])]
            return w.sum() / w.numel()
        if key == "min_action":
            return e.buffers.actions.min()
        if key == "max_action":
            return e.buffers.actions.max()
        raise KeyError(key)

    def consume(self):
        """Snapshot the running means as floats and restart the accumulation (one host sync, used at log time)."""
        out = {k: float(v) for k, v in self.items()}
        self.env.buffers.episode_log.zero_()
        return out


class _CurriculumDistribution(_LazyMapping):
    """`extras["curriculum/distribution"]` (reference legged_robot.py:229-232)."""

    def __init__(self, env):
        self.env = env

    def __iter__(self):
        for c in self.env.category_names:
            yield f"weights_{c}"
            yield f"grid_{c}"

    def __len__(self):
        return 2 * len(self.env.category_names)

    def __getitem__(self, key):
        kind, cat = key.split("_", 1)
        i = self.env.category_names.index(cat)
        self.env.sync_curricula_from_device()
        return self.env.curricula[i].weights if kind == "weights" else self.env.curricula[i].grid


class _SimFaults(_LazyMapping):
    """`extras["sim_faults"]`: occurrences per fault site since the counters were last consumed (include/go1sim.h
    `Go1FaultBit`).  The reference has no such entry — PhysX never returns a non-finite state; here every containment
    of a failed environment is reported instead of hidden.  Lazy: device reads happen on access only."""

    def __init__(self, env):
        self.env = env

    def __iter__(self):
        return iter(H.FAULT_NAMES[b] for b in sorted(H.FAULT_NAMES))

    def __len__(self):
        return len(H.FAULT_NAMES)

    def __getitem__(self, key):
        for b, name in H.FAULT_NAMES.items():
            if name == key:
                return self.env.buffers.fault_counts[b]
        raise KeyError(key)

    def consume(self):
        """Counts as ints (one host sync) and restart; `fatal` = occurrences that ended an episode."""
        counts = self.env.buffers.fault_counts.tolist()
        self.env.buffers.fault_counts.zero_()
        out = {name: int(counts[b]) for b, name in sorted(H.FAULT_NAMES.items())}
        out["fatal"] = sum(int(counts[b]) for b in H.FAULT_NAMES if (H.FAULT_FATAL_MASK >> b) & 1)
        drops = self.env.buffers.contact_drop_counts.tolist()          # contact points that found no solver slot, per class
        self.env.buffers.contact_drop_counts.zero_()
        if any(drops):
            out["contact_dropped_by_class"] = {name: int(drops[c]) for c, name in sorted(H.CONTACT_CLASS_NAMES.items()) if drops[c]}
        return out


class LeggedRobot(BaseTask):
    def __init__(self, cfg, sim_params, physics_engine, sim_device, headless, eval_cfg=None,
                 initial_dynamics_dict=None):
        self.cfg = cfg
        self.eval_cfg = eval_cfg
        self.sim_params = sim_params
        self.height_samples = None
        self.debug_viz = False
        self.init_done = False
        self.initial_dynamics_dict = initial_dynamics_dict
        if eval_cfg is not None:          # reference legged_robot.py:41-42
            self._parse_cfg(eval_cfg)
        self._parse_cfg(cfg)
        super().__init__(cfg, sim_params, physics_engine, sim_device, headless, eval_cfg)
        self._init_buffers()
        self.init_done = True
        self.record_now = False
        self.record_eval_now = False
        self.collecting_evaluation = False
        self.num_still_evaluating = 0

    # ------------------------------------------------------------------------------------------------
    def _parse_cfg(self, cfg):
        """reference legged_robot.py:1716-1732 (derived step counts are written back into cfg, as there)."""
        self.dt = H.policy_dt(cfg)
        self.obs_scales = cfg.obs_scales
        self.curriculum_thresholds = vars(cfg.curriculum_thresholds)
        cfg.command_ranges = vars(cfg.commands)
        if cfg.terrain.mesh_type not in ('heightfield', 'trimesh'):
            cfg.terrain.curriculum = False
        cfg.env.max_episode_length = np.ceil(cfg.env.episode_length_s / self.dt)
        self.max_episode_length = cfg.env.max_episode_length
        dr = cfg.domain_rand
        dr.push_interval = np.ceil(dr.push_interval_s / self.dt)
        dr.rand_interval = np.ceil(dr.rand_interval_s / self.dt)
        dr.gravity_rand_interval = np.ceil(dr.gravity_rand_interval_s / self.dt)
        dr.gravity_rand_duration = np.ceil(dr.gravity_rand_interval * dr.gravity_impulse_duration)

    def create_sim(self):
        """reference legged_robot.py:493-515 + _create_envs :1481-1609: terrain, origins, per-env dynamics
        parameters, then the simulator instance (go1sim_create replaces gym.create_sim / prepare_sim)."""
        cfg = self.cfg
        mesh_type = cfg.terrain.mesh_type
        if mesh_type not in (None, 'plane', 'heightfield', 'trimesh'):
            raise ValueError("Terrain mesh type not recognised. Allowed types are [None, plane, heightfield, trimesh]")
        self.up_axis_idx = 2
        if self.eval_cfg is not None:
            et = self.eval_cfg.terrain
            if et.mesh_type in ('heightfield', 'trimesh') and mesh_type not in ('heightfield', 'trimesh'):
                # (the reference builds no Terrain then, legged_robot.py:500-505, and fails in _get_env_origins)
                raise ValueError("eval_cfg.terrain.mesh_type is a generated terrain but cfg.terrain.mesh_type is not: the "
                                 "evaluation region is appended to the training terrain (terrain.py:37-54)")
            if mesh_type in ('heightfield', 'trimesh') and et.mesh_type in ('heightfield', 'trimesh') and \
                    (et.horizontal_scale, et.vertical_scale) != (cfg.terrain.horizontal_scale, cfg.terrain.vertical_scale):
                raise ValueError("eval_cfg.terrain: horizontal_scale / vertical_scale must equal the training terrain's (both "
                                 "regions live in one height field, converted with the training scales: terrain.py:30-36)")
            if self.num_train_envs % 16 != 0:
                raise ValueError(f"eval_cfg: cfg.env.num_envs = {self.num_train_envs} must be a multiple of 16 (the step kernel "
                                 f"selects the train / evaluation configuration per wavefront of 16 environments)")
        if mesh_type in ('heightfield', 'trimesh'):
            if self.eval_cfg is not None and self.eval_cfg.terrain.mesh_type in ('heightfield', 'trimesh'):
                # reference legged_robot.py:502-503: a second tile grid for the evaluation environments behind the training one
                self.terrain = Terrain(cfg.terrain, self.num_train_envs, self.eval_cfg.terrain, self.num_eval_envs)
            else:
                self.terrain = Terrain(cfg.terrain, self.num_train_envs)
            hs = self.terrain.heightsamples
            self.height_samples = torch.tensor(hs).view(self.terrain.tot_rows, self.terrain.tot_cols).to(self.device)
        seed = int(getattr(cfg, "seed", getattr(cfg.env, "seed", 0)))
        offset = int(getattr(cfg.env, "env_id_offset", 0))
        # Environments sharded over ranks (one process per GPU): the command curriculum stays ONE global curriculum — every
        # rank adds up the per-bin success counts of all shards before the weight update, so weights, CDFs and therefore
        # the sampled commands are those of a single-GPU run over the concatenated shards (RNG streams are
        # keyed by global env id).  one int32 all-reduce of the K x 4 x 441 success counts every K = commands.curriculum_update_interval env steps (7 KB per step at the default K = 1); `Cfg.commands.global_curriculum = False` keeps
        # per-rank curricula instead.
        import torch.distributed as dist
        self._curriculum_sync = bool(dist.is_available() and dist.is_initialized()
                                     and (dist.get_world_size() > 1 or os.environ.get("GO1_FORCE_DP", "0") not in ("", "0"))
                                     and cfg.commands.command_curriculum and getattr(cfg.commands, "global_curriculum", True))
        self.sim_config, self.sim_meta = H.build_sim_config(
            cfg, num_envs=self.num_envs, seed=seed, env_id_offset=offset,
            # solver sweeps per substep = the reference's PhysX setting (num_position_iterations = 4, num_velocity_iterations
            # = 0, legged_robot_config.py:413-414); `num_solver_sweeps` overrides it
            solver_iterations=int(getattr(H.physx_section(cfg), "num_solver_sweeps",
                                          getattr(H.physx_section(cfg), "num_position_iterations", 4))),
            defer_curriculum_update=self._curriculum_sync)
        self.buffers = B = H.SimBuffers(self.sim_config, self.sim_meta, self.device)
        if mesh_type in ('heightfield', 'trimesh'):
            # 'heightfield': the height field's bilinear surface.  'trimesh': the reference corrects faces steeper than slope_treshold into vertical
            # walls (terrain.py:33-36, convert_heightfield_to_trimesh) — simulated as vertical contact faces on the same height
            # field (include/go1sim.h hf_wall_units)
            H.bind_height_field(self.sim_config, B, self.terrain.heightsamples, cfg.terrain.horizontal_scale,
                                cfg.terrain.vertical_scale, cfg.terrain.border_size,
                                slope_threshold=float(getattr(cfg.terrain, "slope_treshold", 0.75)) if mesh_type == 'trimesh' else None)
        self.num_dof = self.num_dofs = self.num_actuated_dof = 12
        self.num_bodies = 17
        self.dof_names = list(H.DOF_NAMES)
        self.body_names = list(H.BODY_NAMES)
        self.feet_indices = torch.tensor([4, 8, 12, 16], device=self.device)
        self.penalised_contact_indices = torch.tensor(
            [i for i in range(17) if self.sim_config.penalised_body_mask >> i & 1], device=self.device)
        self.termination_contact_indices = torch.tensor(
            [i for i in range(17) if self.sim_config.termination_body_mask >> i & 1], device=self.device)
        self._get_env_origins()
        self._init_custom_buffers__()
        self._call_train_eval(self._randomize_rigid_body_props, torch.arange(self.num_envs, device=self.device))
        self.category_names = self.sim_meta["category_names"]
        self.curricula = self.sim_meta["curricula"]
        self.sim = H.Go1Sim(self.sim_config, B, self.sim_device_id)
        self.sim_config_eval = None
        if self.eval_cfg is not None:
            # evaluation environments: the train block with the fields the reference dispatches per group taken from
            # eval_cfg (go1sim_host.EVAL_CFG_FIELDS); selected per wavefront inside the kernels
            S_e, _ = H.build_sim_config(self.eval_cfg, num_envs=self.num_envs, seed=seed, env_id_offset=offset,
                                        solver_iterations=self.sim_config.solver_iterations,
                                        defer_curriculum_update=self._curriculum_sync)
            self.sim_config_eval = H.make_eval_sim_config(self.sim_config, S_e)
            self.sim.set_eval_config(self.sim_config_eval, self.num_train_envs)
        B.episode_sums_eval.fill_(-1.0)          # reference legged_robot.py:1420-1424 (only evaluation environments ever write it)

    def _call_train_eval(self, func, env_ids):
        """reference legged_robot.py:531-544: func(ids, cfg) for the training environments, func(ids, eval_cfg) for the others"""
        env_ids_train = env_ids[env_ids < self.num_train_envs]
        env_ids_eval = env_ids[env_ids >= self.num_train_envs]
        ret, ret_eval = None, None
        if len(env_ids_train) > 0:
            ret = func(env_ids_train, self.cfg)
        if len(env_ids_eval) > 0:
            ret_eval = func(env_ids_eval, self.eval_cfg)
            if ret is not None and ret_eval is not None:
                ret = torch.cat((ret, ret_eval), axis=-1)
        return ret

    def _get_env_origins(self):
        """reference legged_robot.py:1675-1714, once per group (`_call_train_eval(self._get_env_origins, ...)`, :1538): the
        training environments on cfg's tiles or grid, the evaluation environments on eval_cfg's."""
        B, N = self.buffers, self.num_envs
        origins = torch.zeros(N, 3, device=self.device)
        self.terrain_levels = torch.zeros(N, dtype=torch.long, device=self.device)
        self.terrain_types = torch.zeros(N, dtype=torch.long, device=self.device)
        self.custom_origins = False
        for lo_, n_, c_ in ((0, self.num_train_envs, self.cfg), (self.num_train_envs, self.num_eval_envs, self.eval_cfg)):
            if n_ == 0:
                continue
            ter = c_.terrain
            self.custom_origins = ter.mesh_type in ("heightfield", "trimesh")      # (an attribute: the last group's value stays)
            if self.custom_origins:
                lo, hi = (ter.min_init_terrain_level, ter.max_init_terrain_level) if ter.curriculum else (0, ter.num_rows - 1)
                if ter.center_robots:
                    lo, hi = ter.num_rows // 2 - ter.center_span, ter.num_rows // 2 + ter.center_span - 1
                    tlo, thi = ter.num_cols // 2 - ter.center_span, ter.num_cols // 2 + ter.center_span - 1
                    levels = torch.randint(lo, hi + 1, (n_,), device=self.device)
                    types = torch.randint(tlo, thi + 1, (n_,), device=self.device)
                else:
                    levels = torch.randint(lo, hi + 1, (n_,), device=self.device)
                    types = torch.div(torch.arange(n_, device=self.device), (n_ / ter.num_cols), rounding_mode='floor').to(torch.long)
                ter.max_terrain_level = ter.num_rows
                ter.terrain_origins = torch.from_numpy(ter.env_origins).to(self.device).to(torch.float)
                self.terrain_levels[lo_:lo_ + n_], self.terrain_types[lo_:lo_ + n_] = levels, types
                origins[lo_:lo_ + n_] = ter.terrain_origins[levels, types]
            else:
                cols = np.floor(np.sqrt(n_))
                rows = np.ceil(n_ / cols)
                xx, yy = torch.meshgrid(torch.arange(rows), torch.arange(cols), indexing="ij")
                origins[lo_:lo_ + n_, 0] = c_.env.env_spacing * xx.flatten()[:n_].to(self.device)
                origins[lo_:lo_ + n_, 1] = c_.env.env_spacing * yy.flatten()[:n_].to(self.device)
        # `_reset_root_states` branches on this attribute for every environment (reference :966-980), whatever its group
        self.sim_config.custom_origins = int(self.custom_origins)
        B.env_origins.copy_(origins.t())
        self.env_origins = B.env_origins.t()

    def _init_custom_buffers__(self):
        """reference legged_robot.py:1260-1297: views of the per-env dynamics parameters (+ optional presets)."""
        B = self.buffers
        B.friction_coeffs.fill_(1.0)          # rigid_shape_props_asset[1].friction: Isaac Gym default material
        B.restitutions.fill_(0.0)
        self.default_friction, self.default_restitution = 1.0, 0.0
        self.friction_coeffs = B.friction_coeffs.unsqueeze(1).expand(-1, 4)
        self.restitutions = B.restitutions.unsqueeze(1).expand(-1, 4)
        self.payloads = B.payloads
        self.com_displacements = B.com_displacements.t()
        self.motor_strengths = B.motor_strengths.t()
        self.motor_offsets = B.motor_offsets.t()
        self.Kp_factors = B.Kp_factors.t()
        self.Kd_factors = B.Kd_factors.t()
        if self.initial_dynamics_dict is not None:
            for k, v in self.initial_dynamics_dict.items():
                v = v.to(self.device)
                if k in ("friction_coeffs", "restitutions"):
                    getattr(B, k).copy_(v.reshape(self.num_envs, -1)[:, 0])
                elif k == "payloads":
                    B.payloads.copy_(v)
                elif k in ("com_displacements", "motor_strengths", "Kp_factors", "Kd_factors"):
                    getattr(B, k).copy_(v.t())

    def _randomize_rigid_body_props(self, env_ids, cfg):
        """reference legged_robot.py:611-633 (set-up time; the step kernel re-draws nothing here unless
        randomize_rigids_after_start, which is not on the train.py path)."""
        B, dr, n = self.buffers, cfg.domain_rand, len(env_ids)
        u = lambda rng, *shape: torch.rand(*shape, device=self.device) * (rng[1] - rng[0]) + rng[0]
        # (values preset through `initial_dynamics_dict` are overwritten by this draw when the switch of their quantity is on, as in
        # the reference: _init_custom_buffers__ :1283-1288 runs before it, :1547-1548)
        if dr.randomize_base_mass:
            B.payloads[env_ids] = u(dr.added_mass_range, n)
        if dr.randomize_com_displacement:
            B.com_displacements[:, env_ids] = u(dr.com_displacement_range, 3, n)
        if dr.randomize_friction:
            B.friction_coeffs[env_ids] = u(dr.friction_range, n)
        if dr.randomize_restitution:
            B.restitutions[env_ids] = u(dr.restitution_range, n)

    def _init_buffers(self):
        """Expose the simulator's SoA buffers under the reference's attribute names (legged_robot.py:1123-1258).
        `.t()` views keep the reference's (N, ...) shapes and in-place write semantics."""
        B, N = self.buffers, self.num_envs
        S = self.sim_config
        self.root_states = B.root_states.t()
        self.base_pos = self.root_states[:, 0:3]
        self.base_quat = self.root_states[:, 3:7]
        self.dof_pos = B.dof_pos.t()
        self.dof_vel = B.dof_vel.t()
        self.contact_forces = B.contact_forces.view(17, 3, N).permute(2, 0, 1)
        self.foot_positions = B.foot_positions.view(4, 3, N).permute(2, 0, 1)
        self.foot_velocities = B.foot_velocities.view(4, 3, N).permute(2, 0, 1)
        self.prev_foot_velocities = B.prev_foot_velocities.view(4, 3, N).permute(2, 0, 1)
        self.base_lin_vel = B.base_lin_vel.t()
        self.base_ang_vel = B.base_ang_vel.t()
        self.projected_gravity = B.projected_gravity.t()
        self.torques = B.torques.t()
        self.actions = B.actions.t()
        self.last_actions = B.last_actions.t()
        self.last_last_actions = B.last_last_actions.t()
        self.joint_pos_target = B.joint_pos_target.t()
        self.last_joint_pos_target = B.last_joint_pos_target.t()
        self.last_last_joint_pos_target = B.last_last_joint_pos_target.t()
        self.last_dof_vel = B.last_dof_vel.t()
        # what the actuator network carries from substep to substep (reference :1226-1231)
        self.joint_pos_err_last = B.joint_pos_err_last.t()
        self.joint_pos_err_last_last = B.joint_pos_err_last_last.t()
        self.joint_vel_last = B.joint_vel_last.t()
        self.joint_vel_last_last = B.joint_vel_last_last.t()
        self.p_gains = torch.full((12,), float(S.kp), device=self.device)                 # :1206-1222 (one gain for every joint)
        self.d_gains = torch.full((12,), float(S.kd), device=self.device)
        self.base_init_state = torch.tensor(list(S.base_init_state), device=self.device)  # :1590-1594
        self.forward_vec = torch.tensor([1.0, 0.0, 0.0], device=self.device).repeat(N, 1)
        self.commands = B.commands.t()[:, :self.cfg.commands.num_commands]
        self.gait_indices = B.gait_indices
        self.clock_inputs = B.clock_inputs.t()
        self.desired_contact_states = B.desired_contact_states.t()
        self.foot_indices = B.foot_indices.t()
        self.episode_length_buf = B.episode_length_buf
        self.reset_buf = B.reset_buf
        self.time_out_buf = B.time_out_buf.view(torch.bool)
        self.last_contacts = B.last_contacts.t().view(torch.bool)
        self.rew_buf = B.rew_buf
        self.obs_buf = B.obs_buf
        self.privileged_obs_buf = B.privileged_obs_buf[:, :self.num_privileged_obs]
        self.default_dof_pos = torch.tensor(list(S.default_dof_pos), device=self.device).unsqueeze(0)
        self.torque_limits = torch.tensor(list(S.torque_limits), device=self.device)
        self.dof_pos_limits = torch.tensor([list(S.dof_pos_soft_lower), list(S.dof_pos_soft_upper)], device=self.device).t()
        self.noise_scale_vec = torch.tensor(list(S.noise_scale_vec)[:self.num_obs], device=self.device)
        self.commands_scale = torch.tensor(list(S.commands_scale)[:self.cfg.commands.num_commands], device=self.device)
        self.reward_names = list(self.sim_meta["reward_names"])
        self.reward_scales = dict(self.sim_meta["reward_scales"])
        self.episode_sum_names = list(self.sim_meta["episode_sum_names"])
        self.episode_sums = {n: B.episode_sums[i] for i, n in enumerate(self.episode_sum_names)}
        self.command_sums = {n: B.command_sums[i] for i, n in enumerate(self.sim_meta["command_sum_names"])}
        self.common_step_counter = 0
        self.measured_heights = B.measured_heights.t() if S.measure_heights else 0
        self.add_noise = self.cfg.noise.add_noise
        nt = self.num_train_envs
        self.extras = {"env_bins": B.env_command_bins[:nt], "train/episode": _EpisodeStats(self), "sim_faults": _SimFaults(self)}
        self.fault_flags = B.fault_flags
        # evaluation environments (reference legged_robot.py:188-195, 1420-1424): their first finished episode's sums;
        # the reference's extras["eval/episode"] is created empty and never filled — kept so for the Runner's logging branch
        self.episode_sums_eval = {n: B.episode_sums_eval[i] for i, n in enumerate(self.episode_sum_names)}
        if self.num_eval_envs > 0:
            self.extras["eval/episode"] = {}
        if self.cfg.env.send_timeouts:
            self.extras["time_outs"] = self.time_out_buf[:nt]
        if self.cfg.commands.command_curriculum:
            self.extras["curriculum/distribution"] = _CurriculumDistribution(self)

    # ------------------------------------------------------------------------------------------------
    def step(self, actions):
        """reference legged_robot.py:60-88 — one kernel launch, no host sync."""
        actions = actions.to(device=self.device, dtype=torch.float32)
        if not actions.is_contiguous():
            actions = actions.contiguous()
        self.sim.step(actions)
        if self._curriculum_sync and (self.common_step_counter + 1) % self.sim_config.curriculum_update_interval == 0:
            # ONE exchange for the last `curriculum_update_interval` steps' success counts (a slot per step), then the per-step
            # updates in order: every rank applies what a single process over the concatenated shards would
            import torch.distributed as dist
            dist.all_reduce(self.buffers.curriculum_success)
            self.sim.curriculum_update()
        self.common_step_counter += 1
        return self.obs_buf, self.privileged_obs_buf, self.rew_buf, self.reset_buf, self.extras

    # gravity is a function of (seed, step counter) that kernels and host evaluate alike: no device buffer, no read-back
    @property
    def gravities(self):
        """(N, 3) offset of the gravity vector from (0, 0, -9.8) now in force (reference `self.gravities`, :549-555)"""
        g = torch.from_numpy(H.gravity_at(self.sim_config, self.common_step_counter) - np.array([0.0, 0.0, -9.8], np.float32))
        return g.to(self.device).unsqueeze(0).repeat(self.num_envs, 1)

    @property
    def gravity_vec(self):
        """(N, 3) unit vector along the gravity now in force (reference `self.gravity_vec`, :559)"""
        g = torch.from_numpy(H.gravity_at(self.sim_config, self.common_step_counter))
        return (g / g.norm()).to(self.device).unsqueeze(0).repeat(self.num_envs, 1)

    default_body_mass = 4.801          # base link after the fixed-joint collapse (trunk + imu; csrc/go1_model_data.h GO1_BODY_MASS[0])

    def reset_idx(self, env_ids):
        """reference legged_robot.py:150-239."""
        if len(env_ids) == 0:
            return
        if len(env_ids) == self.num_envs:
            self.sim.reset_idx(None)
        else:
            self.sim.reset_idx(torch.as_tensor(env_ids, device=self.device))

    def set_idx_pose(self, env_ids, dof_pos, base_state):
        """reference legged_robot.py:241-261: the buffers ARE the simulator state, so writing them is the push."""
        if len(env_ids) == 0:
            return
        if dof_pos is not None:
            self.dof_pos[env_ids] = dof_pos.to(self.device)
            self.dof_vel[env_ids] = 0.
        self.root_states[env_ids] = base_state.to(self.device)

    def set_main_agent_pose(self, loc, quat):
        self.root_states[0, 0:3] = torch.tensor(loc, device=self.device, dtype=torch.float)
        self.root_states[0, 3:7] = torch.tensor(quat, device=self.device, dtype=torch.float)

    @property
    def env_command_bins(self):
        return self.buffers.env_command_bins.cpu().numpy()

    @property
    def env_command_categories(self):
        return self.buffers.env_command_categories.cpu().numpy()

    def sync_curricula_from_device(self):
        w = self.buffers.curriculum_weights.double().cpu().numpy()
        for cur, row in zip(self.curricula, w):
            cur.weights = row

    def sync_curricula_to_device(self):
        """After a host-side edit of `curricula[i].weights` (e.g. Runner resume, ppo_cse/__init__.py:83-91)."""
        w = np.stack([c.weights for c in self.curricula]).astype(np.float32)
        self.buffers.curriculum_weights.copy_(torch.from_numpy(w))
        cdf = np.cumsum(w.astype(np.float64), axis=1)
        self.buffers.curriculum_cdf.copy_(torch.from_numpy((cdf / cdf[:, -1:]).astype(np.float32)))

    # ---- recording: no renderer on this stack; the Runner calls these every iteration ------------------
    def start_recording(self):
        self.record_now = True

    def start_recording_eval(self):
        self.record_eval_now = True

    def pause_recording(self):
        self.record_now = False

    def pause_recording_eval(self):
        self.record_eval_now = False

    def get_complete_frames(self):
        return []

    def get_complete_frames_eval(self):
        return []

    def render(self, mode="rgb_array"):
        raise NotImplementedError("no renderer on the MI355X stack (SURVEY.md §5 video: out of scope)")
