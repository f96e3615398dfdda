"""Height-field terrain: the height scan against the reference's own `_get_heights` (golden, CPU oracle here and HIP in
the gpu test), terrain tile generation, and physics invariants on an inclined height field."""
import os

import numpy as np
import pytest
import torch

import go1sim_host as H
from util import GOLDEN, make_sim, standing_state


HEIGHT_FIXTURES = ["heights.npz", "heights_coarse.npz"]      # 187 points on a 0.1 m grid; 15 points, 0.25 m grid, 5 m border


def heights_sim(N, extra=None, fname="heights.npz"):
    d = np.load(os.path.join(GOLDEN, fname))
    ex = {"terrain": dict(measure_heights=True, measured_points_x=[float(x) for x in d["points_x"]],
                          measured_points_y=[float(y) for y in d["points_y"]])}
    for k, v in (extra or {}).items():
        ex.setdefault(k, {}).update(v)
    cfg, S, meta, B = make_sim("train", N, extra=ex)
    H.bind_height_field(S, B, d["height_samples"], float(d["hscale"]), float(d["vscale"]), float(d["border"]))
    B.root_states[:] = torch.from_numpy(d["root_states"]).t()
    return d, cfg, S, meta, B


@pytest.mark.parametrize("fname", HEIGHT_FIXTURES)
def test_height_scan_matches_reference(oracle_lib, fname):
    d, cfg, S, meta, B = heights_sim(len(np.load(os.path.join(GOLDEN, fname))["root_states"]), fname=fname)
    assert S.terrain_type == 1 and S.measure_heights == 1 and S.num_height_x * S.num_height_y == d["heights"].shape[1]
    B.commands[4] = 3.0
    B.commands[8] = 0.5
    orc = oracle_lib.Oracle(S, B)
    orc.post_physics(np.array([0.0, 0.0, -9.8]))
    np.testing.assert_allclose(B.measured_heights.t().numpy(), d["heights"], rtol=0, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("fname", HEIGHT_FIXTURES)
def test_hip_height_scan_matches_reference(fname):
    d, cfg, S, meta, Bc = heights_sim(len(np.load(os.path.join(GOLDEN, fname))["root_states"]), fname=fname)
    Bc.commands[4] = 3.0
    Bc.commands[8] = 0.5
    Bg = Bc.clone_to("cuda:0")
    sim = H.Go1Sim(S, Bg, 0)
    sim.post_physics([0.0, 0.0, -9.8])
    torch.cuda.synchronize()
    np.testing.assert_allclose(Bg.measured_heights.t().cpu().numpy(), d["heights"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("fname", HEIGHT_FIXTURES)
def test_emulated_height_scan_matches_reference(fname):
    """the kernel's own scan code (csrc/go1_maps.h) on the CPU through the SIMT emulator"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "emu"))
    import emu_sim
    d, cfg, S, meta, B = heights_sim(len(np.load(os.path.join(GOLDEN, fname))["root_states"]), fname=fname)
    B.commands[4] = 3.0
    B.commands[8] = 0.5
    sim = emu_sim.EmuSim(S, B)
    sim.post_physics([0.0, 0.0, -9.8])
    np.testing.assert_allclose(B.measured_heights.t().numpy(), d["heights"], rtol=0, atol=1e-6)


def test_terrain_layout_matches_reference():
    """tile grid, borders, the evaluation region behind the training one, `make_terrain`'s mapping and the env origins against
    the reference `Terrain` class (terrain.py:12-179) run over the same sub-terrain generators (terrain_layout.npz)."""
    import json
    import types
    from util import GOLDEN
    from go1_gym.utils import terrain as T
    d = np.load(os.path.join(GOLDEN, "terrain_layout.npz"))
    conf = json.loads(str(d["config"]))
    for tag, with_eval in (("solo", False), ("split", True)):
        tr, ev = types.SimpleNamespace(**conf["train"]), types.SimpleNamespace(**conf["eval"])
        np.random.seed(int(d["seed"]))
        ter = T.Terrain(tr, 32, ev, 16) if with_eval else T.Terrain(tr, 32)
        assert (ter.tot_rows, ter.tot_cols) == tuple(d[f"{tag}_tot"])
        assert np.array_equal(ter.heightsamples, d[f"{tag}_heights"]) and ter.heightsamples.dtype == np.int16
        np.testing.assert_array_equal(tr.env_origins, d[f"{tag}_train_origins"])
        assert (tr.x_offset, tr.rows_offset) == (0, 0)
        if with_eval:
            np.testing.assert_array_equal(ev.env_origins, d[f"{tag}_eval_origins"])
            assert (ev.x_offset, ev.rows_offset) == tuple(d[f"{tag}_eval_offsets"])
            assert ev.env_origins[..., 0].min() > tr.terrain_length * tr.num_rows          # behind the training region
            assert d[f"{tag}_heights"][tr.tot_rows:].any()


def test_terrain_generators_and_tile_grid():
    from go1_gym.envs.base.legged_robot_config import make_cfg
    from go1_gym.utils import terrain as T
    cfg = make_cfg()
    cfg.terrain.num_rows, cfg.terrain.num_cols, cfg.terrain.border_size = 3, 10, 5
    np.random.seed(0)
    ter = T.Terrain(cfg.terrain, 8)
    px = int(cfg.terrain.terrain_length / cfg.terrain.horizontal_scale)
    b = int(5 / cfg.terrain.horizontal_scale)
    assert ter.heightsamples.shape == (3 * px + 2 * b, 10 * px + 2 * b) and ter.heightsamples.dtype == np.int16
    assert (ter.heightsamples[:b] == 0).all() and (ter.heightsamples[:, :b] == 0).all()          # flat border
    assert cfg.terrain.env_origins.shape == (3, 10, 3)
    assert cfg.terrain.env_origins[1, 0, 0] == pytest.approx(1.5 * cfg.terrain.terrain_length)
    # difficulty grows with the row (curriculum mode): the stairs column gets taller
    stairs_col = 6      # choice 0.601: ascending stairs (0.55 <= choice < 0.8)
    assert cfg.terrain.env_origins[2, stairs_col, 2] > cfg.terrain.env_origins[0, stairs_col, 2] > 0
    tile = T.SubTerrain(width=80, length=80, vertical_scale=0.005, horizontal_scale=0.1)
    T.pyramid_stairs_terrain(tile, step_width=0.31, step_height=0.1, platform_size=3.0)
    levels = np.unique(tile.height_field_raw)
    assert len(levels) > 3 and np.all(np.diff(levels) == 20)                                   # 0.1 m risers
    tile = T.SubTerrain(width=80, length=80, vertical_scale=0.005, horizontal_scale=0.1)
    T.pyramid_sloped_terrain(tile, slope=0.2, platform_size=2.0)
    g = np.diff(tile.height_field_raw[:20, 40].astype(float)) * 0.005 / 0.1      # below the platform clip
    assert g.mean() == pytest.approx(0.2, rel=0.1)
    tile = T.SubTerrain(width=80, length=80, vertical_scale=0.005, horizontal_scale=0.1)
    T.random_uniform_terrain(tile, -0.05, 0.05, step=0.005, downsampled_scale=0.2, rng=np.random.default_rng(0))
    assert abs(tile.height_field_raw).max() <= 10 and tile.height_field_raw.std() > 1
    # train.py's terrain is flat
    from scripts.train_config import apply_train_config
    flat = T.Terrain(apply_train_config(make_cfg()).terrain, 8)
    assert flat.heightsamples.shape == (1500, 1500) and not flat.heightsamples.any()


def inclined_field(rows, cols, slope, hscale=0.1, vscale=0.0005):
    x = np.arange(rows)[:, None] * hscale
    return np.broadcast_to(np.rint(slope * x / vscale), (rows, cols)).astype(np.int16), hscale, vscale


@pytest.mark.parametrize("slope,mu_robot,holds", [(0.1, 2.0, True), (0.2, 0.0, False)])
def test_incline_friction_holds_or_slides(oracle_lib, slope, mu_robot, holds):
    """Robot placed parallel to an incline.  With friction (pair mu 1.5 > tan(theta)) the feet stay put and the net
    contact force balances gravity: (0, 0, m g) although every contact normal is tilted.  Without friction it slides
    downhill at the analytic g sin(theta) and the contact force is purely along the slope normal."""
    N = 4
    cfg, S, meta, B = make_sim("alt", N, extra={"domain_rand": dict(randomize_gravity=False)})
    hs, hscale, vscale = inclined_field(400, 60, slope)
    H.bind_height_field(S, B, hs, hscale, vscale, 0.0)
    if not holds:
        S.terrain_friction = S.terrain_dynamic_friction = 0.0
    standing_state(S, B, z=0.30)
    th = np.arctan(slope)
    B.root_states[0] = 20.0
    B.root_states[1] = 3.0
    B.root_states[2] = 0.30 / np.cos(th) + slope * 20.0
    B.root_states[4] = -np.sin(th / 2)          # pitch the trunk onto the slope (nose up towards +x)
    B.root_states[6] = np.cos(th / 2)
    B.friction_coeffs[:] = mu_robot
    orc = oracle_lib.Oracle(S, B)
    a = np.zeros((N, 12), np.float32)
    mg = 11.309932 * 9.8
    if holds:
        for _ in range(125):
            orc.step(a)
        x0 = B.foot_positions.view(4, 3, N)[:, 0].clone()      # feet, not the base: the soft-PD legs let the body sway
        F = 0
        for _ in range(25):
            orc.step(a)
            F = F + B.contact_forces.view(17, 3, N).sum(0) / 25
        dx = (B.foot_positions.view(4, 3, N)[:, 0] - x0).numpy()
        assert float(B.reset_buf.sum()) == 0
        assert np.abs(dx).max() < 0.005
        np.testing.assert_allclose(F[2].numpy(), mg, rtol=0.02)
        np.testing.assert_allclose(F[:2].numpy(), 0, atol=0.03 * mg)
    else:
        for _ in range(25):
            orc.step(a)
        v0 = B.root_states[7:10].clone()
        for _ in range(50):
            orc.step(a)
        acc = ((B.root_states[7:10] - v0) / (50 * 0.02)).numpy()          # world frame, m/s^2
        g_par = 9.8 * np.sin(th)
        np.testing.assert_allclose(acc[0], -g_par * np.cos(th), rtol=0.02)
        np.testing.assert_allclose(acc[2], -g_par * np.sin(th), rtol=0.2)       # + residual heave of the soft legs
        F = B.contact_forces.view(17, 3, N).sum(0).numpy()
        np.testing.assert_allclose(F[0] / F[2], -slope, atol=0.01)       # along the slope normal (-slope, 0, 1)/|.|
        np.testing.assert_allclose(np.linalg.norm(F, axis=0), mg * np.cos(th), rtol=0.03)


def test_heights_above_terrain_makes_the_reward_independent_of_the_ground_elevation(oracle_lib):
    """`rewards.heights_above_terrain` (NOT a reference switch; include/go1sim.h reward_heights_above_terrain): the reference's
    _reward_feet_clearance_cmd_linear / _reward_feet_contact_vel / _reward_jump read world z (corl_rewards.py:129 `# - reference_heights`,
    :100, :52-53), so lifting the whole scene by 1 m changes the reward; with the switch the same robots on the lifted ground get the
    reward they got at z = 0 — which is what a terrain-curriculum tile grid needs (BASELINE configs[2])."""
    import pyoracle
    N = 16
    rews = {}
    for above in (False, True):
        for lift in (0.0, 1.0):
            ex = {"domain_rand": dict(randomize_gravity=False), "rewards": dict(heights_above_terrain=above)}
            cfg, S, meta, B = make_sim("train_noise", N, seed=3, extra=ex)
            hs = np.full((240, 240), int(round(lift / 0.005)), dtype=np.int16)
            H.bind_height_field(S, B, hs, 0.1, 0.005, 0.0)
            B.env_origins[0].uniform_(5.0, 18.0, generator=torch.Generator().manual_seed(1))
            B.env_origins[1].uniform_(5.0, 18.0, generator=torch.Generator().manual_seed(2))
            B.env_origins[2] = lift
            orc = pyoracle.Oracle(S, B)
            orc.reset_idx()
            rng = np.random.default_rng(0)
            tot = np.zeros(N)
            for _ in range(5):
                orc.step((0.3 * rng.standard_normal((N, 12))).astype(np.float32))
                tot += B.rew_buf.numpy()
            rews[(above, lift)] = tot
    np.testing.assert_allclose(rews[(True, 1.0)], rews[(True, 0.0)], rtol=1e-4, atol=1e-6)       # (fp32 state at z ~ 1.3 instead of 0.3)
    np.testing.assert_allclose(rews[(True, 0.0)], rews[(False, 0.0)], rtol=1e-6, atol=1e-8)      # on the z = 0 ground the switch changes nothing
    assert np.abs(rews[(False, 1.0)] - rews[(False, 0.0)]).max() > 1e-3                           # the reference terms do depend on the elevation
