"""Base class of the vectorised task: device selection, VecEnv attribute surface, reset()
(mirror of reference go1_gym/envs/base/base_task.py:14-137; no viewer on this stack)."""
import gym
import torch


def parse_device_str(device_str):
    parts = str(device_str).split(":")
    return parts[0], int(parts[1]) if len(parts) > 1 else 0


class BaseTask(gym.Env):
    def __init__(self, cfg, sim_params, physics_engine, sim_device, headless, eval_cfg=None):
        self.sim_params = sim_params
        self.physics_engine = physics_engine
        self.sim_device = sim_device
        self.headless = headless          # accepted and ignored: no viewer
        self.device = self._resolve_device(sim_device)
        self.graphics_device_id = -1
        self.num_obs = cfg.env.num_observations
        self.num_privileged_obs = cfg.env.num_privileged_obs
        self.num_actions = cfg.env.num_actions
        if eval_cfg is not None:          # reference base_task.py:43-49: evaluation environments behind the training ones
            self.num_eval_envs = eval_cfg.env.num_envs
            self.num_train_envs = cfg.env.num_envs
            self.num_envs = self.num_eval_envs + self.num_train_envs
        else:
            self.num_eval_envs = 0
            self.num_train_envs = cfg.env.num_envs
            self.num_envs = cfg.env.num_envs
        self.extras = {}
        self.viewer = None
        self.enable_viewer_sync = True
        self.create_sim()

    def _resolve_device(self, sim_device):
        """'cuda:N' (HIP device N under PyTorch-ROCm) -> torch device string of every buffer.  The product has no other
        case: the Go1 step exists only as HIP kernels."""
        sim_device_type, self.sim_device_id = parse_device_str(sim_device)
        if sim_device_type != "cuda":
            raise RuntimeError(
                f"sim_device={sim_device!r}: the Go1 step runs only as HIP kernels on an MI355X ('cuda:N' under "
                f"PyTorch-ROCm); there is no CPU simulation path in the product")
        return f"cuda:{self.sim_device_id}"

    def get_observations(self):
        return self.obs_buf

    def get_privileged_observations(self):
        return self.privileged_obs_buf

    def reset_idx(self, env_ids):
        raise NotImplementedError

    def reset(self):
        self.reset_idx(torch.arange(self.num_envs, device=self.device))
        obs, privileged_obs, _, _, _ = self.step(
            torch.zeros(self.num_envs, self.num_actions, device=self.device, requires_grad=False))
        return obs, privileged_obs

    def step(self, actions):
        raise NotImplementedError

    def render_gui(self, sync_frame_time=True):
        pass

    def close(self):
        pass
