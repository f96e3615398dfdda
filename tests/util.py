"""Shared helpers for the tests: build configs/buffers, load golden fixtures into SoA buffers."""
import os

import numpy as np
import torch

import go1sim_host as H
from go1_gym.envs.base.legged_robot_config import make_cfg
from scripts.train_config import apply_train_config
from golden.variants import apply_variant

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def make_cfg_variant(variant="train", num_envs=16, extra=None):
    cfg = apply_train_config(make_cfg(), num_envs=num_envs)
    if variant != "train_noise":
        apply_variant(cfg, variant)
    for section, values in (extra or {}).items():
        for k, v in values.items():
            setattr(getattr(cfg, section), k, v)
    return cfg


def make_sim(variant="train", num_envs=16, seed=3, extra=None, **kw):
    cfg = make_cfg_variant(variant, num_envs, extra)
    S, meta = H.build_sim_config(cfg, seed=seed, **kw)
    B = H.SimBuffers(S, meta, "cpu")
    return cfg, S, meta, B


def standing_state(S, B, z=0.30):
    B.root_states.zero_()
    B.root_states[2] = z
    B.root_states[6] = 1.0
    B.dof_pos[:] = torch.tensor(list(S.default_dof_pos)).unsqueeze(1)
    B.dof_vel.zero_()


def randomize_dr(B, seed=0):
    g = torch.Generator().manual_seed(seed)
    B.friction_coeffs.uniform_(0.1, 3.0, generator=g)
    B.restitutions.uniform_(0.0, 0.4, generator=g)
    B.payloads.uniform_(-1.0, 3.0, generator=g)


def load_maps_fixture(name, S, meta, B):
    """Fill SoA buffers from a tests/golden/maps_*.npz fixture (reference layouts are (N, ...))."""
    d = np.load(os.path.join(GOLDEN, name))
    t = lambda k: torch.from_numpy(np.ascontiguousarray(d[k]))
    N = S.num_envs
    B.root_states[:] = t("root_states").t()
    B.dof_pos[:] = t("dof_pos").t()
    B.dof_vel[:] = t("dof_vel").t()
    B.foot_positions[:] = t("foot_positions").reshape(N, 12).t()
    B.foot_velocities[:] = t("foot_velocities").reshape(N, 12).t()
    B.prev_foot_velocities[:] = t("prev_foot_velocities").reshape(N, 12).t()
    B.contact_forces[:] = t("contact_forces").reshape(N, 51).t()
    for k in ("actions", "last_actions", "last_last_actions", "joint_pos_target", "last_joint_pos_target",
              "last_last_joint_pos_target", "last_dof_vel", "torques", "motor_strengths", "motor_offsets"):
        getattr(B, k)[:] = t(k).t()
    B.last_contacts[:] = t("last_contacts").t().to(torch.uint8)
    B.commands[:] = t("commands").t()
    B.gait_indices[:] = t("gait_indices")
    B.episode_length_buf[:] = (t("episode_length_buf") - 1).to(torch.int32)     # the oracle increments first
    for k in ("friction_coeffs", "restitutions", "payloads"):
        getattr(B, k)[:] = t(k)
    B.com_displacements[:] = t("com_displacements").t()
    es_names = [str(x) for x in d["episode_sum_names"]]
    cs_names = [str(x) for x in d["command_sum_names"]]
    assert es_names == meta["episode_sum_names"], (es_names, meta["episode_sum_names"])
    assert cs_names == meta["command_sum_names"]
    B.episode_sums[:] = t("episode_sums")
    B.command_sums[:] = t("command_sums")
    return d


def maps_fixture_stream(fname):
    """(simulator seed, step counter before the call) a maps fixture was generated for: the noise fixture fixes both (its
    `torch.rand_like` was fed the uniforms of that stream), the others carry no draw of their own."""
    d = np.load(os.path.join(GOLDEN, fname))
    return (int(d["sim_seed"]), int(d["step"]) - 1) if "sim_seed" in d.files else (3, 7)


def maps_keep(d, S):
    """environments of a maps fixture whose outputs are comparable one to one: not reset (re-initialised from the simulator's own
    stream) and not on the DOF-property re-randomisation cadence this step (the fixture runs the reference's maps, not its
    `_post_physics_step_callback`, so the privileged observation there still shows the old motor strengths / offsets)."""
    keep = ~d["out_reset_buf"].astype(bool)
    return keep & (d["episode_length_buf"] % int(S.rand_interval) != 0)


RESAMPLE_MODES = {"gaitwise": dict(gaitwise_curricula=True, exclusive_phase_offset=False, balance_gait_distribution=False, binary_phases=True),
                  "exclusive": dict(gaitwise_curricula=False, exclusive_phase_offset=True, balance_gait_distribution=False, binary_phases=True),
                  "balance": dict(gaitwise_curricula=False, exclusive_phase_offset=False, balance_gait_distribution=True, binary_phases=False),
                  "plain": dict(gaitwise_curricula=False, exclusive_phase_offset=False, balance_gait_distribution=False, binary_phases=False),
                  # the same four branches with the other setting of `binary_phases` (legged_robot.py:814-817, 1361)
                  "gaitwise_smooth": dict(gaitwise_curricula=True, exclusive_phase_offset=False, balance_gait_distribution=False, binary_phases=False),
                  "exclusive_smooth": dict(gaitwise_curricula=False, exclusive_phase_offset=True, balance_gait_distribution=False, binary_phases=False),
                  "balance_binary": dict(gaitwise_curricula=False, exclusive_phase_offset=False, balance_gait_distribution=True, binary_phases=True),
                  "plain_binary": dict(gaitwise_curricula=False, exclusive_phase_offset=False, balance_gait_distribution=False, binary_phases=True)}


def load_resample_fixture(mode):
    """tests/golden/resample_<mode>.npz (reference `_resample_commands` + curriculum update, make_golden.py gen_resample) ->
    (fixture, S, meta, B) with the pre-resample state in CPU buffers."""
    d = np.load(os.path.join(GOLDEN, f"resample_{mode}.npz"))
    N = d["commands0"].shape[0]
    cfg, S, meta, B = make_sim("train", N, seed=int(d["sim_seed"]), extra={"commands": RESAMPLE_MODES[mode]})
    assert meta["category_names"] == [str(x) for x in d["category_names"]]
    assert meta["command_sum_names"] == [str(x) for x in d["command_sum_names"]]
    B.commands[:] = torch.from_numpy(d["commands0"]).t()
    B.command_sums[:] = torch.from_numpy(d["command_sums0"])
    B.env_command_bins[:] = torch.from_numpy(d["bins0"]).int()
    B.env_command_categories[:] = torch.from_numpy(d["cats0"]).int()
    w0 = d["weights0"].astype(np.float32)
    B.curriculum_weights[:] = torch.from_numpy(w0)
    cdf = np.cumsum(w0.astype(np.float64), axis=1)
    B.curriculum_cdf[:] = torch.from_numpy((cdf / cdf[:, -1:]).astype(np.float32))
    return d, S, meta, B


def check_resample_against_reference(d, B, atol=1e-6):
    """commands / bins / categories / cleared sums of the resampled envs, untouched rows of the others, then the weights."""
    ids = d["env_ids"]
    rest = np.setdiff1d(np.arange(B.commands.shape[1]), ids)
    cmd = B.commands.t().cpu().numpy()
    np.testing.assert_array_equal(B.env_command_categories.cpu().numpy()[ids], d["cats1"][ids])
    np.testing.assert_array_equal(B.env_command_bins.cpu().numpy()[ids], d["bins1"][ids])
    np.testing.assert_allclose(cmd[ids], d["commands1"][ids], rtol=0, atol=atol)
    np.testing.assert_array_equal(cmd[rest], d["commands0"][rest])
    sums = B.command_sums.cpu().numpy()
    assert np.all(sums[:, ids] == 0) and np.array_equal(sums[:, rest], d["command_sums0"][:, rest])
    np.testing.assert_allclose(B.curriculum_weights.cpu().numpy(), d["weights1"], rtol=0, atol=1e-6)
    assert np.abs(d["weights1"] - d["weights0"]).sum() > 1.0


def check_exported_policy_layout(ckpt_dir):
    """the Runner's TorchScript exports against the structure of the reference's shipped `adaptation_module_latest.jit`
    (tests/golden/pretrain_jit_layout.json): same parameter names / shapes / dtype and module sequence, loadable on the CPU the
    way scripts/play.py:17-29 loads them, and composable as body(cat(history, adaptation_module(history)))."""
    import json
    with open(os.path.join(GOLDEN, "pretrain_jit_layout.json")) as f:
        ref = json.load(f)
    adapt = torch.jit.load(os.path.join(ckpt_dir, "adaptation_module_latest.jit"))
    body = torch.jit.load(os.path.join(ckpt_dir, "body_latest.jit"))
    assert adapt.original_name == ref["original_name"] == body.original_name
    assert {k: list(v.shape) for k, v in adapt.state_dict().items()} == ref["state_dict"]
    assert [[n, c.original_name] for n, c in adapt.named_children()] == ref["children"]
    assert sorted({str(v.dtype) for v in adapt.state_dict().values()}) == ref["dtypes"]
    assert all(v.device.type == "cpu" for v in list(adapt.state_dict().values()) + list(body.state_dict().values()))
    assert [c.original_name for _, c in body.named_children()] == ["Linear", "ELU", "Linear", "ELU", "Linear", "ELU", "Linear"]
    assert {k: tuple(v.shape) for k, v in body.state_dict().items()}["0.weight"] == (512, 2102)
    hist = torch.randn(3, 2100)
    assert body(torch.cat((hist, adapt(hist)), dim=-1)).shape == (3, 12)
    weights = torch.load(os.path.join(ckpt_dir, "ac_weights_last.pt"), map_location="cpu")
    groups = {k.split(".")[0] for k in weights}
    assert groups == {"std", "adaptation_module", "actor_body", "critic_body"}, groups      # ppo_cse/__init__.py:231-251 consumers


def self_pair_codes(w2):
    """contact signature word 2 (include/go1sim.h): bits 3 p .. 3 p + 2 = 1 + type of the leg-leg self-contact listed for pair p of
    (0,1) (0,2) (0,3) (1,2) (1,3) (2,3) — type 0 lower-lower, 1 lower-thigh, 2 thigh-lower, 3 thigh-thigh, 4 hip-lower, 5 lower-hip; 0: none —,
    bits 18..21: lower leg of leg 0..3 against the trunk, bits 28..31: legs with limit rows.  Returns (six codes, trunk mask)."""
    w2 = int(w2) & 0xFFFFFFFF
    return [(w2 >> (3 * p)) & 7 for p in range(6)], (w2 >> 18) & 0xF


def self_contacts_listed(w2):
    codes, trunk = self_pair_codes(w2)
    return sum(c != 0 for c in codes) + bin(trunk).count("1")
