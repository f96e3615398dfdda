"""TEST INFRASTRUCTURE: stands in for `go1sim_host.Go1Sim` where there is no GPU (the authoring container), so that the
host-side classes — LeggedRobot, VelocityTrackingEasyEnv, HistoryWrapper, Runner, the reference's own scripts — can be
executed end to end.  It steps CPU `SimBuffers` with the fp64 oracle (oracle/ is the checker; only tests may use it).
Never importable from the product: it lives under tests/ and is installed by monkeypatching."""
import numpy as np
import torch

import pyoracle


class OracleBackedSim:
    def __init__(self, S, buffers, device_index=0):
        assert buffers.device.type == "cpu"
        self.S, self.buffers = S, buffers
        self.orc = pyoracle.Oracle(S, buffers)
        self._timing = 0

    def step(self, actions):
        self.orc.step(actions.detach().cpu().numpy())

    def reset_idx(self, ids=None):
        self.orc.reset_idx(None if ids is None else ids.detach().cpu().numpy().astype(np.int32))

    def append_history(self):
        S, B, c = self.S, self.buffers, self.orc.ctr
        no, R = S.num_obs, S.num_obs_history + 1
        B.obs_history[:, c.history_slot * no:(c.history_slot + 1) * no] = B.obs_buf
        B.obs_history[:, (c.history_slot + R) * no:(c.history_slot + R + 1) * no] = B.obs_buf
        c.history_slot = (c.history_slot + 1) % R

    def history_window_offset(self):
        return ((self.orc.ctr.history_slot + 1) % (self.S.num_obs_history + 1)) * self.S.num_obs

    def set_config(self, S):
        self.S = self.orc.S = S

    def set_eval_config(self, S_eval, num_train_envs):
        self.orc.set_eval_config(S_eval, num_train_envs)

    def counters(self):
        return self.orc.ctr.common_step_counter, self.orc.ctr.lag_head

    def set_counters(self, counter, lag_head):
        self.orc.ctr.common_step_counter, self.orc.ctr.lag_head = int(counter), int(lag_head)

    def enable_timing(self, capacity):
        self._timing = capacity

    def read_timings(self, max_n=65536):
        return []

    def curriculum_update(self):
        self.orc.curriculum_update()


def install(monkeypatch):
    """CPU buffers + oracle stepping behind the real env classes (device check and library load bypassed)."""
    import go1sim_host as H
    from go1_gym.envs.base import base_task

    def resolve(self, sim_device):
        self.sim_device_id = 0
        return "cpu"
    monkeypatch.setattr(base_task.BaseTask, "_resolve_device", resolve)
    monkeypatch.setattr(H, "Go1Sim", OracleBackedSim)
