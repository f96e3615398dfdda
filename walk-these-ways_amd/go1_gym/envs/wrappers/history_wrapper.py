"""`HistoryWrapper` (mirror of reference go1_gym/envs/wrappers/history_wrapper.py:6-41): dict observations with
the rolling (N, H*num_obs) observation history.

The reference rebuilds the history with `torch.cat` every step (reads and writes N x 2100 floats, :23).  Here the
step kernel appends the 70 new floats to a double-length ring in HBM and `obs_history` is a strided *view* of the
current window — same values, same (oldest first) order, no copy.  Quirks kept: the history is not cleared on
per-env resets (:32-35 is never called by the env) and `get_observations()` appends the current obs once more
(:29), which `Runner.learn` relies on right after `reset()`."""
import gym
import torch


class HistoryWrapper(gym.Wrapper):
    def __init__(self, env):
        super().__init__(env)
        self.env = env
        self.obs_history_length = self.env.cfg.env.num_observation_history
        self.num_obs_history = self.obs_history_length * self.num_obs
        self.num_privileged_obs = self.num_privileged_obs

    @property
    def obs_history(self):
        off = self.env.sim.history_window_offset()
        return self.env.buffers.obs_history[:, off:off + self.num_obs_history]

    def step(self, action):
        obs, rew, done, info = self.env.step(action)
        return {'obs': obs, 'privileged_obs': info["privileged_obs"], 'obs_history': self.obs_history}, rew, done, info

    def get_observations(self):
        obs = self.env.get_observations()
        privileged_obs = self.env.get_privileged_observations()
        self.env.sim.append_history()
        return {'obs': obs, 'privileged_obs': privileged_obs, 'obs_history': self.obs_history}

    def reset_idx(self, env_ids):
        ret = self.env.reset_idx(env_ids)
        self.env.buffers.obs_history[env_ids, :] = 0
        return ret

    def reset(self):
        ret = self.env.reset()
        privileged_obs = self.env.get_privileged_observations()
        self.env.buffers.obs_history.zero_()
        return {"obs": ret, "privileged_obs": privileged_obs, "obs_history": self.obs_history}
