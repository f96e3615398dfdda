"""Command resampling and the curriculum update against the REFERENCE's own code (tests/golden/resample_*.npz:
`LeggedRobot._resample_commands` + `RewardThresholdCurriculum.update` executed by make_golden.py with the RNG consumers fed
the Philox uniforms of the same (seed, env, step)): success criterion, +0.2 neighbourhood update, category assignment,
bin -> command decode, the gait remaps of all four modes (gaitwise, exclusive_phase_offset, balance_gait_distribution, none),
binary phases, small-command zeroing, command-sum reset.  CPU: the C oracle; GPU: the HIP kernels through the C-ABI."""
import ctypes

import pytest
import torch

import go1sim_host as H
from util import RESAMPLE_MODES, check_resample_against_reference, load_resample_fixture


@pytest.mark.parametrize("mode", list(RESAMPLE_MODES))
def test_oracle_resample_and_curriculum_update_match_reference(oracle_lib, mode):
    d, S, meta, B = load_resample_fixture(mode)
    U = d["uniforms"]                       # the fixture's uniforms are the oracle's own stream
    L = oracle_lib.lib()
    for e in (0, 17, 63):
        for j in (0, 1, 9, 17):
            assert L.go1_oracle_uniform(ctypes.byref(S), e, int(d["step"]), int(d["purpose"]), j) == U[e, j]
    orc = oracle_lib.Oracle(S, B)
    orc.ctr.common_step_counter = int(d["step"])
    orc.reset_idx(d["env_ids"])
    orc.curriculum_update()
    check_resample_against_reference(d, B)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", list(RESAMPLE_MODES))
def test_hip_resample_and_curriculum_update_match_reference(mode):
    d, S, meta, Bc = load_resample_fixture(mode)
    Bg = Bc.clone_to("cuda:0")
    sim = H.Go1Sim(S, Bg, 0)
    sim.set_counters(int(d["step"]), 0)
    sim.reset_idx(torch.from_numpy(d["env_ids"]))
    sim.curriculum_update()
    torch.cuda.synchronize()
    check_resample_against_reference(d, Bg)


def _load_reset_fixture():
    import os
    import numpy as np
    from util import GOLDEN, make_sim
    d = np.load(os.path.join(GOLDEN, "reset.npz"))
    N = d["dof_pos0"].shape[0]
    cfg, S, meta, B = make_sim("alt", N, seed=int(d["sim_seed"]))
    t = lambda k: torch.from_numpy(d[k])
    B.dof_pos[:] = t("dof_pos0").t(); B.dof_vel[:] = t("dof_vel0").t(); B.root_states[:] = t("root_states0").t()
    B.env_origins[:] = t("env_origins").t()
    for k in ("motor_strengths", "motor_offsets", "Kp_factors", "Kd_factors"):
        getattr(B, k)[:] = t(k + "0").t()
    B.last_actions.fill_(3.0); B.last_last_actions.fill_(3.0); B.last_dof_vel.fill_(3.0); B.lag_buffer.fill_(3.0)
    B.gait_indices.fill_(0.3); B.episode_length_buf.fill_(57)
    return d, S, meta, B


def _check_reset(d, B):
    import numpy as np
    ids = d["env_ids"]
    rest = np.setdiff1d(np.arange(B.dof_pos.shape[1]), ids)
    g = lambda k: B.tensors[k].cpu().numpy()
    # fp32 on both sides, same uniforms; the two sides associate (hi-lo)*u+lo differently: a few ulp
    np.testing.assert_allclose(g("dof_pos").T, d["dof_pos1"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(g("dof_vel").T, d["dof_vel1"], rtol=0, atol=0)
    np.testing.assert_allclose(g("root_states").T, d["root_states1"], rtol=1e-6, atol=2e-6)
    for k in ("motor_strengths", "motor_offsets", "Kp_factors", "Kd_factors"):
        np.testing.assert_allclose(g(k).T, d[k + "1"], rtol=1e-6, atol=1e-7, err_msg=k)
    assert np.abs(d["root_states1"][ids] - d["root_states0"][ids]).max() > 0.1
    for k in ("last_actions", "last_last_actions", "last_dof_vel"):                 # legged_robot.py:173-178
        assert np.all(g(k)[:, ids] == 0) and np.all(g(k)[:, rest] == 3.0), k
    assert np.all(g("lag_buffer")[:, :, ids] == 0) and np.all(g("lag_buffer")[:, :, rest] == 3.0)      # :236-239
    assert np.all(g("gait_indices")[ids] == 0) and np.all(g("episode_length_buf")[ids] == 0) and np.all(g("reset_buf")[ids] == 1)
    assert np.all(g("episode_length_buf")[rest] == 57)


def test_oracle_reset_matches_reference(oracle_lib):
    """reset distribution (DOF properties, joint angles, root pose / velocity) against the reference's own
    `_randomize_dof_props` / `_reset_dofs` / `_reset_root_states` fed the same uniforms (tests/golden/reset.npz)."""
    d, S, meta, B = _load_reset_fixture()
    orc = oracle_lib.Oracle(S, B)
    orc.ctr.common_step_counter = int(d["step"])
    orc.reset_idx(d["env_ids"])
    _check_reset(d, B)


@pytest.mark.gpu
def test_hip_reset_matches_reference():
    d, S, meta, Bc = _load_reset_fixture()
    Bg = Bc.clone_to("cuda:0")
    sim = H.Go1Sim(S, Bg, 0)
    sim.set_counters(int(d["step"]), 0)
    sim.reset_idx(torch.from_numpy(d["env_ids"]))
    torch.cuda.synchronize()
    _check_reset(d, Bg)
