"""`VelocityTrackingEasyEnv` (mirror of reference go1_gym/envs/go1/velocity_tracking/__init__.py:10-49):
the 4-tuple gym-style step on top of `LeggedRobot`, with the per-step diagnostic arrays in `extras`.

The reference copies 11 tensors to host numpy every step (11 blocking D2H syncs, :28-42).  Here the same keys
are served by a lazy mapping: a key is transferred only when somebody reads it, so `step()` stays sync-free."""
import numpy as np
import torch

from go1_gym.envs.base.legged_robot import LeggedRobot
from go1_gym.envs.base.legged_robot_config import Cfg


class SimParams:
    """Plain container standing in for gymapi.SimParams (only `dt`/`substeps`/... are read on this stack)."""

    def __init__(self, sim_cfg):
        for k, v in sim_cfg.items():
            setattr(self, k, v)
        self.dt = float(np.float32(self.dt))      # SimParams.dt is float32 in Isaac Gym


class _LazyExtras(dict):
    """dict whose registered producers are evaluated on item access."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self._lazy = {}

    def set_lazy(self, key, fn):
        self._lazy[key] = fn
        dict.__setitem__(self, key, None)

    def __getitem__(self, key):
        fn = self._lazy.get(key)
        return fn() if fn is not None else dict.__getitem__(self, key)

    def get(self, key, default=None):
        return self[key] if key in self else default

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]

    def __setitem__(self, key, value):
        self._lazy.pop(key, None)
        dict.__setitem__(self, key, value)


class VelocityTrackingEasyEnv(LeggedRobot):
    def __init__(self, sim_device, headless, num_envs=None, prone=False, deploy=False,
                 cfg: Cfg = None, eval_cfg: Cfg = None, initial_dynamics_dict=None, physics_engine="SIM_PHYSX"):
        if num_envs is not None:
            cfg.env.num_envs = num_envs
        sim_params = SimParams(vars(cfg.sim))
        super().__init__(cfg, sim_params, physics_engine, sim_device, headless, eval_cfg, initial_dynamics_dict)
        ex = _LazyExtras(self.extras)
        npy = lambda t: (lambda: t.detach().cpu().numpy().copy())
        ex["privileged_obs"] = self.privileged_obs_buf
        ex.set_lazy("joint_pos", npy(self.dof_pos))
        ex.set_lazy("joint_vel", npy(self.dof_vel))
        ex.set_lazy("joint_pos_target", npy(self.joint_pos_target))
        ex["joint_vel_target"] = torch.zeros(12)
        ex.set_lazy("body_linear_vel", npy(self.base_lin_vel))
        ex.set_lazy("body_angular_vel", npy(self.base_ang_vel))
        ex.set_lazy("body_linear_vel_cmd", lambda: self.commands.cpu().numpy()[:, 0:2])
        ex.set_lazy("body_angular_vel_cmd", lambda: self.commands.cpu().numpy()[:, 2:])
        ex.set_lazy("contact_states", lambda: (self.contact_forces[:, self.feet_indices, 2] > 1.).cpu().numpy().copy())
        ex.set_lazy("foot_positions", npy(self.foot_positions))
        ex.set_lazy("body_pos", npy(self.root_states[:, 0:3]))
        ex.set_lazy("torques", npy(self.torques))
        self.extras = ex

    def step(self, actions):
        obs, _, rew, done, extras = super().step(actions)
        return obs, rew, done, extras

    def reset(self):
        self.reset_idx(torch.arange(self.num_envs, device=self.device))
        obs, _, _, _ = self.step(torch.zeros(self.num_envs, self.num_actions, device=self.device, requires_grad=False))
        return obs
