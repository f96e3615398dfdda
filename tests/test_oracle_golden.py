"""Pin the CPU oracle's tensor maps against the reference's own Python (golden fixtures produced by
tests/golden/make_golden.py from /root/reference).  Tolerances: the reference
computes in fp32, the oracle in fp64 -> 1e-5 relative on sums, 1e-6 absolute on elementwise maps
(erf/sin/exp chains 1e-5)."""
import os

import numpy as np
import pytest
import torch

import go1sim_host as H
from util import GOLDEN, load_maps_fixture, make_sim, maps_fixture_stream, maps_keep
from golden.variants import FUZZ_VARIANTS


@pytest.mark.parametrize("variant,fname", [("train", "maps_train.npz"), ("train", "maps_train_mild.npz"),
                                           ("alt", "maps_alt.npz"), ("alt", "maps_alt_mild.npz"), ("alt2", "maps_alt2.npz"),
                                           ("alt2", "maps_alt2_mild.npz"), ("train_noise", "maps_train_noise_mild.npz")]
                                          + [(f"fuzz{k}", f"maps_fuzz{k}_mild.npz") for k in range(FUZZ_VARIANTS)])
def test_post_physics_maps_match_reference(oracle_lib, variant, fname):
    N = 48
    seed, counter = maps_fixture_stream(fname)          # (the noise fixture: observation noise from the Philox stream, :375-376)
    cfg, S, meta, B = make_sim(variant, N, seed=seed)
    d = load_maps_fixture(fname, S, meta, B)
    assert [str(x) for x in d["reward_names"]] == meta["reward_names"]
    np.testing.assert_allclose(d["reward_scales"], [meta["reward_scales"][n] for n in meta["reward_names"]], rtol=1e-6)
    assert int(d["out_max_episode_length"]) == S.max_episode_length
    np.testing.assert_allclose(d["out_noise_scale_vec"], np.array(list(S.noise_scale_vec))[:S.num_obs], rtol=1e-6)
    np.testing.assert_allclose(d["out_dof_pos_soft_limits"][:, 0], list(S.dof_pos_soft_lower), rtol=1e-6)
    np.testing.assert_allclose(d["out_dof_pos_soft_limits"][:, 1], list(S.dof_pos_soft_upper), rtol=1e-6)
    orc = oracle_lib.Oracle(S, B)
    orc.ctr.common_step_counter = counter
    orc.post_physics(d["gravity"].astype(np.float64))

    reset = d["out_reset_buf"].astype(bool)
    np.testing.assert_array_equal(B.reset_buf.numpy().astype(bool), reset)
    np.testing.assert_array_equal(B.time_out_buf.numpy().astype(bool), d["out_time_out_buf"].astype(bool))
    assert reset.any() and (~reset).any()
    keep = maps_keep(d, S)
    tol = dict(rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(B.base_lin_vel.t().numpy(), d["out_base_lin_vel"], **tol)
    np.testing.assert_allclose(B.base_ang_vel.t().numpy(), d["out_base_ang_vel"], **tol)
    np.testing.assert_allclose(B.projected_gravity.t().numpy(), d["out_projected_gravity"], **tol)
    np.testing.assert_allclose(B.gait_indices.numpy()[keep], d["out_gait_indices"][keep], **tol)
    np.testing.assert_allclose(B.foot_indices.t().numpy(), d["out_foot_indices"], **tol)
    np.testing.assert_allclose(B.clock_inputs.t().numpy(), d["out_clock_inputs"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(B.desired_contact_states.t().numpy(), d["out_desired_contact_states"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(B.rew_buf.numpy(), d["out_rew_buf"], rtol=1e-4, atol=1e-7)
    assert np.abs(d["out_rew_buf"]).max() > 0 or "mild" not in fname
    np.testing.assert_array_equal(B.last_contacts.t().numpy().astype(bool), d["out_last_contacts"].astype(bool))
    np.testing.assert_allclose(B.episode_sums.numpy()[:, keep], d["out_episode_sums"][:, keep], rtol=1e-4, atol=1e-5)
    # command sums are cleared for reset envs by the resample; compare the others
    np.testing.assert_allclose(B.command_sums.numpy()[:, keep], d["out_command_sums"][:, keep], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(B.obs_buf.numpy()[keep], d["out_obs"][keep], rtol=1e-5, atol=2e-6)
    if variant == "train_noise":
        assert S.add_noise and np.count_nonzero(d["out_noise_scale_vec"]) == 3 + 12 + 12          # gravity, joint angles, joint rates
    np.testing.assert_allclose(B.privileged_obs_buf.numpy()[keep][:, :S.num_privileged_obs], d["out_priv"][keep], rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("variant", ["train", "alt", "act_nolag", "pd_lag"])
def test_compute_torques_matches_reference(oracle_lib, variant):
    d = np.load(os.path.join(GOLDEN, f"torques_{variant}.npz"))
    steps, N = d["actions"].shape[:2]
    cfg, S, meta, B = make_sim(variant, N)
    for k in ("motor_strengths", "motor_offsets", "Kp_factors", "Kd_factors"):
        getattr(B, k)[:] = torch.from_numpy(d[k]).t()
    orc = oracle_lib.Oracle(S, B)
    for s in range(steps):
        B.dof_pos[:] = torch.from_numpy(d["dof_pos"][s]).t()
        B.dof_vel[:] = torch.from_numpy(d["dof_vel"][s]).t()
        orc.compute_torques(np.ascontiguousarray(d["actions"][s].T))
        np.testing.assert_allclose(B.joint_pos_target.t().numpy(), d["joint_pos_target"][s], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(B.torques.t().numpy(), d["torques"][s], rtol=1e-5, atol=2e-5)
    if variant == "alt":
        assert np.abs(d["torques"]).max() > 33.4     # the clip at 33.5 is exercised by the PD branch


def test_actuator_net_known_answer(oracle_lib):
    # zero input -> -0.0041 N m (verified by loading the TorchScript file)
    assert abs(oracle_lib.actuator_net(np.zeros((1, 6)))[0] - (-0.0041)) < 5e-5


def test_philox_known_answers(oracle_lib):
    # Random123 kat_vectors, philox4x32-10
    kat = [([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
            [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for ctr, key, out in kat:
        assert [int(x) for x in oracle_lib.philox(ctr, key)] == out


def test_curriculum_matches_reference():
    from go1_gym.envs.base.curriculum import RewardThresholdCurriculum
    d = np.load(os.path.join(GOLDEN, "curriculum.npz"))
    kw = dict(x_vel=(-5.0, 5.0, 21), y_vel=(-0.6, 0.6, 1), yaw_vel=(-5.0, 5.0, 21), body_height=(-0.25, 0.15, 2),
              gait_frequency=(2.0, 4.0, 3))
    c = RewardThresholdCurriculum(seed=100, **kw)
    np.testing.assert_allclose(c.grid, d["grid"], rtol=0, atol=1e-12)
    c.set_to(low=d["low"], high=d["high"])
    np.testing.assert_array_equal(c.weights, d["weights0"])
    c.update(d["bins"], [torch.from_numpy(d["rew0"]), torch.from_numpy(d["rew1"])], [0.5, 0.5], local_range=d["local_range"])
    np.testing.assert_allclose(c.weights, d["weights1"], atol=1e-12)
    np.testing.assert_array_equal(c.get_local_bins(np.array([0, 300, 1322]), ranges=d["local_range"]), d["local"])
    samples, inds = c.sample(64)      # same RandomState(100) stream as the reference class
    np.testing.assert_array_equal(inds, d["inds"])
    np.testing.assert_allclose(samples, d["samples"], atol=1e-12)
    ptr, idx = c.neighbourhood_csr(d["local_range"])
    for b, row in zip([0, 300, 1322], d["local"]):
        np.testing.assert_array_equal(idx[ptr[b]:ptr[b + 1]], row.nonzero()[0])


def test_sum_curriculum_matches_reference():
    """`SumCurriculum`, `is_met`, `key_is_met` (reference curriculum.py:6-14, 92-109; sum_curriculum.npz)"""
    from go1_gym.envs.base.curriculum import SumCurriculum, is_met, key_is_met
    d = np.load(os.path.join(GOLDEN, "sum_curriculum.npz"))
    c = SumCurriculum(seed=3, x=(-1.0, 1.0, 5), y=(0.0, 2.0, 3), z=(0.0, 1.0, 2))
    for b, e in zip(d["bins"], d["errs"]):
        c.update(b, e, 0.4)
    assert np.array_equal(c.success, d["success"]) and np.array_equal(c.trials, d["trials"])
    np.testing.assert_array_equal(c.success_rates("x", "y", "z"), d["rates_all"])
    np.testing.assert_allclose(c.success_rates("x"), d["rates_x"], rtol=1e-15)
    np.testing.assert_allclose(c.success_rates("x", "z"), d["rates_xz"], rtol=1e-15)
    assert [bool(is_met(2.0, 0.5, 0.3)), bool(is_met(2.0, 0.7, 0.3)), bool(key_is_met(None, None, 10, "k", 0, 0.1))] == list(d["met"])
