"""What the runners expect of an environment (the contract of reference go1_gym_learn/env/vec_env.py:10-40).

`LeggedRobot` / `HistoryWrapper` satisfy it by duck typing, as in the reference (nothing there derives from the class
either), so `isinstance(env, VecEnv)` is answered structurally: any object with the four calls and the attributes passes."""
import abc

import torch

_SIZES = ("num_envs", "num_obs", "num_privileged_obs", "num_actions", "max_episode_length")
_TENSORS = ("obs_buf", "privileged_obs_buf", "rew_buf", "reset_buf", "episode_length_buf")      # the last: steps since reset
_CALLS = ("step", "reset", "get_observations", "get_privileged_observations")


class VecEnv(abc.ABC):
    __annotations__ = {**{n: int for n in _SIZES}, **{n: torch.Tensor for n in _TENSORS}, "extras": dict, "device": torch.device}

    @abc.abstractmethod
    def step(self, actions):
        """actions (num_envs, num_actions) -> (obs, privileged_obs or None, rewards, dones, extras)"""

    @abc.abstractmethod
    def reset(self, env_ids):
        """re-initialise the listed environments"""

    @abc.abstractmethod
    def get_observations(self):
        """-> obs_buf"""

    @abc.abstractmethod
    def get_privileged_observations(self):
        """-> privileged_obs_buf, or None when the task has none"""

    @classmethod
    def __subclasshook__(cls, other):
        if cls is VecEnv:
            return all(callable(getattr(other, c, None)) for c in _CALLS) or NotImplemented
        return NotImplemented
