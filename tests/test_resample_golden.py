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


# ---- the train / evaluation dispatch (`_call_train_eval`, legged_robot.py:531-544) against the reference ---------------------
EVAL_OVERRIDES = {"domain_rand": dict(motor_strength_range=[1.3, 1.5], motor_offset_range=[0.05, 0.08], Kp_factor_range=[1.4, 1.6],
                                      Kd_factor_range=[0.2, 0.4]),
                  "terrain": dict(x_init_range=0.2, y_init_range=0.3, yaw_init_range=0.4, x_init_offset=1.5, y_init_offset=-2.5)}       # = make_golden.py


def _load_reset_eval_fixture():
    import os
    import numpy as np
    from util import GOLDEN, make_sim
    d = np.load(os.path.join(GOLDEN, "reset_eval.npz"))
    N = d["dof_pos0"].shape[0]
    cfg, S, meta, B = make_sim("alt", N, seed=int(d["sim_seed"]))
    _, S_full, _, _ = make_sim("alt", N, seed=int(d["sim_seed"]), extra=EVAL_OVERRIDES)
    S_eval = H.make_eval_sim_config(S, S_full)
    t = lambda k: torch.from_numpy(d[k])
    B.dof_pos[:] = t("dof_pos0").t(); B.dof_vel[:] = t("dof_vel0").t(); B.root_states[:] = t("root_states0").t()
    B.env_origins[:] = t("env_origins").t()
    for k in ("motor_strengths", "motor_offsets", "Kp_factors", "Kd_factors"):
        getattr(B, k)[:] = t(k + "0").t()
    B.last_actions.fill_(3.0); B.last_last_actions.fill_(3.0); B.last_dof_vel.fill_(3.0); B.lag_buffer.fill_(3.0)
    B.gait_indices.fill_(0.3); B.episode_length_buf.fill_(57)
    return d, S, S_eval, int(d["num_train"]), B


def _check_reset_eval(d, B, NT):
    import numpy as np
    _check_reset(d, B)
    ids = d["env_ids"]
    ev = ids[ids >= NT]
    ms = B.motor_strengths.cpu().numpy().T
    assert ms[ev].min() >= 1.3 and ms[ev].max() <= 1.5 and ms[ids[ids < NT]].max() < 1.3          # each group from ITS range


def test_oracle_train_eval_reset_matches_reference(oracle_lib):
    """`_call_train_eval` around `_randomize_dof_props` / `_reset_dofs` / `_reset_root_states` executed by the reference's own
    code with a second configuration for the evaluation environments (tests/golden/reset_eval.npz) against the oracle's
    per-environment configuration selection."""
    d, S, S_eval, NT, B = _load_reset_eval_fixture()
    orc = oracle_lib.Oracle(S, B)
    orc.set_eval_config(S_eval, NT)
    orc.ctr.common_step_counter = int(d["step"])
    orc.reset_idx(d["env_ids"])
    _check_reset_eval(d, B, NT)
    orc.S_eval = None
    orc._apply_eval()                        # (the split is module-global in the oracle: switch it off for the tests that follow)


def test_emulated_train_eval_reset_matches_reference():
    """the same fixture through the product's reset kernel (go1_env_kernel: configuration selected per lane), executed by
    the SIMT emulator"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "emu"))
    import emu_sim
    d, S, S_eval, NT, B = _load_reset_eval_fixture()
    sim = emu_sim.EmuSim(S, B)
    sim.set_eval_config(S_eval, NT)
    sim.set_counters(int(d["step"]), 0)
    sim.reset_idx(torch.from_numpy(d["env_ids"]).to(torch.int32))
    _check_reset_eval(d, B, NT)


@pytest.mark.gpu
def test_hip_train_eval_reset_matches_reference():
    d, S, S_eval, NT, Bc = _load_reset_eval_fixture()
    Bg = Bc.clone_to("cuda:0")
    sim = H.Go1Sim(S, Bg, 0)
    sim.set_eval_config(S_eval, NT)
    sim.set_counters(int(d["step"]), 0)
    sim.reset_idx(torch.from_numpy(d["env_ids"]))
    torch.cuda.synchronize()
    _check_reset_eval(d, Bg, NT)


# ---- the randomising branches of _post_physics_step_callback against the reference (tests/golden/callbacks.npz) -------------
CALLBACK_FIXTURES = ["callbacks.npz", "callbacks_offset.npz"]       # the second: the teleport window of an evaluation region


def _load_callbacks_fixture(fname):
    import os
    import numpy as np
    from util import GOLDEN, load_maps_fixture, make_sim
    d0 = np.load(os.path.join(GOLDEN, fname))
    N = d0["root_states"].shape[0]
    cfg, S, meta, B = make_sim("dr", N, seed=int(d0["sim_seed"]))
    S.teleport_x_offset = float(d0["teleport_x_offset"])          # int(cfg.terrain.x_offset * horizontal_scale), :1033
    d = load_maps_fixture(fname, S, meta, B)
    B.Kp_factors[:] = torch.from_numpy(d["Kp_factors"]).t()
    B.Kd_factors[:] = torch.from_numpy(d["Kd_factors"]).t()
    assert S.push_robots and S.teleport_robots and S.randomize_rigids_after_start
    return d, S, B


def _check_callbacks(d, B):
    import numpy as np
    g = lambda k: B.tensors[k].cpu().numpy()
    keep = g("reset_buf") == 0                    # (an environment that terminated was re-initialised after the callback)
    assert keep.sum() >= 0.8 * keep.size
    rs0, rs1 = d["root_states"], d["out_root_states"]
    got = g("root_states").T
    np.testing.assert_allclose(got[keep][:, 0:2], rs1[keep][:, 0:2], rtol=1e-6, atol=1e-5)          # teleport (:1028-1051)
    np.testing.assert_allclose(got[keep][:, 7:9], rs1[keep][:, 7:9], rtol=1e-6, atol=1e-6)          # push (:1017-1026)
    assert (np.abs(rs1[:, 0:2] - rs0[:, 0:2]).max(1) > 1.0).sum() > 10 and len(d["push_ids"]) > 10 and len(d["rand_ids"]) > 10
    for k in ("motor_strengths", "motor_offsets", "Kp_factors", "Kd_factors", "com_displacements"):
        np.testing.assert_allclose(g(k).T[keep], d["out_" + k][keep], rtol=1e-6, atol=1e-7, err_msg=k)
    for k in ("payloads", "friction_coeffs", "restitutions"):
        np.testing.assert_allclose(g(k)[keep], d["out_" + k][keep], rtol=1e-6, atol=1e-7, err_msg=k)
    changed = np.abs(d["out_payloads"] - d["payloads"]) > 0
    assert changed[d["rand_ids"]].all() and not changed[np.setdiff1d(np.arange(len(changed)), d["rand_ids"])].any()


@pytest.mark.parametrize("fname", CALLBACK_FIXTURES)
def test_oracle_callbacks_match_reference(oracle_lib, fname):
    """`_teleport_robots`, `_push_robots`, `_randomize_dof_props`, `_randomize_rigid_body_props` on their episode-length
    cadence (legged_robot.py:675-708) — the reference's own methods fed the oracle's Philox uniforms (callbacks.npz)."""
    d, S, B = _load_callbacks_fixture(fname)
    orc = oracle_lib.Oracle(S, B)
    orc.ctr.common_step_counter = int(d["step"]) - 1
    orc.post_physics(d["gravity"].astype("float64"))
    _check_callbacks(d, B)


@pytest.mark.parametrize("fname", CALLBACK_FIXTURES)
def test_emulated_callbacks_match_reference(fname):
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "emu"))
    import emu_sim
    d, S, B = _load_callbacks_fixture(fname)
    sim = emu_sim.EmuSim(S, B)
    sim.set_counters(int(d["step"]) - 1, 0)
    sim.post_physics(d["gravity"])
    _check_callbacks(d, B)


@pytest.mark.gpu
@pytest.mark.parametrize("fname", CALLBACK_FIXTURES)
def test_hip_callbacks_match_reference(fname):
    d, S, Bc = _load_callbacks_fixture(fname)
    Bg = Bc.clone_to("cuda:0")
    sim = H.Go1Sim(S, Bg, 0)
    sim.set_counters(int(d["step"]) - 1, 0)
    sim.post_physics(d["gravity"])
    torch.cuda.synchronize()
    _check_callbacks(d, Bg)


def test_oracle_gravity_schedule_matches_reference(oracle_lib):
    """`_randomize_gravity` on its cadence (legged_robot.py:546-561, :701-705, :1549): the vector in force during the policy
    step with pre-increment counter t over three intervals — the impulse is up for `duration` steps, then exactly the
    nominal gravity — reference statements executed by make_golden.py gen_gravity."""
    import os
    import numpy as np
    from util import GOLDEN, make_sim
    d = np.load(os.path.join(GOLDEN, "gravity.npz"))
    cfg, S, meta, B = make_sim("dr", 16, seed=int(d["sim_seed"]))
    assert (S.gravity_rand_interval, S.gravity_rand_duration) == (int(d["interval"]), int(d["duration"]))
    orc = oracle_lib.Oracle(S, B)
    got = np.stack([orc.gravity_at(t) for t in range(len(d["gravity"]))])
    np.testing.assert_allclose(got, d["gravity"], rtol=0, atol=2e-6)
    quiet = np.abs(d["gravity"][:, :2]).max(1) == 0
    assert quiet.sum() == 3 * (int(d["interval"]) - int(d["duration"])) and (got[quiet] == np.array([0.0, 0.0, -9.8], np.float32).astype(np.float64)).all()
    unit = got / np.linalg.norm(got, axis=1, keepdims=True)                  # gravity_vec (:559), what projected_gravity rotates
    np.testing.assert_allclose(unit, d["gravity_vec"], atol=1e-6)


def test_host_gravity_schedule_matches_reference_and_oracle(oracle_lib):
    """`go1sim_host.gravity_at` (what `env.gravities` / `env.gravity_vec` report) against the reference's schedule (gravity.npz)
    and against the oracle under a second seed and range."""
    import os
    import numpy as np
    from util import GOLDEN, make_sim
    d = np.load(os.path.join(GOLDEN, "gravity.npz"))
    cfg, S, meta, B = make_sim("dr", 16, seed=int(d["sim_seed"]))
    got = np.stack([H.gravity_at(S, t) for t in range(len(d["gravity"]))])
    np.testing.assert_allclose(got, d["gravity"], rtol=0, atol=2e-6)
    cfg, S, meta, B = make_sim("dr", 16, seed=(7 << 32) + 12345, extra={"domain_rand": dict(gravity_range=[-2.0, 0.5])})
    orc = oracle_lib.Oracle(S, B)
    for t in (0, 1, 396, 397, 400, 401, 802, 5000, 123456):
        np.testing.assert_allclose(H.gravity_at(S, t), orc.gravity_at(t), rtol=0, atol=2e-6)
