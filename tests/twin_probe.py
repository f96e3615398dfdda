"""Determinism / twin probe of the step kernel's height-field instances (test infrastructure, run by hand on the GPU box:
`python tests/twin_probe.py [variant.so ...]`).  Three simulators per library on the relief of test_gpu_parity.run_height_field_comparison,
4096 environments x 40 free-running steps: the product instance twice (A, B) and its `_sig` twin (T), stepped from the same state with the
same actions and re-aligned onto A after every step.  A vs B differing = the kernel is not deterministic (a race); A == B but A vs T
differing = the two template instances were compiled to different arithmetic."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
PKG = os.path.join(REPO, "walk-these-ways_amd")
for p in (os.path.join(PKG, "shims"), PKG, os.path.join(REPO, "oracle"), REPO, HERE):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import go1sim_host as H  # noqa: E402
import test_gpu_parity as T  # noqa: E402

KEYS = ["root_states", "dof_pos", "dof_vel", "rew_buf", "torques", "foot_positions", "measured_heights", "obs_buf", "contact_forces", "reset_buf"]


def align(dst, src):
    for k, t in src.tensors.items():
        if t is not None and dst.tensors.get(k) is not None and k != "contact_signature":
            dst.tensors[k].copy_(t)


def probe(walls, N=4096, steps=40):
    import pyoracle
    pts_x = [round(-0.8 + 0.1 * i, 1) for i in range(17)]
    pts_y = [round(-0.5 + 0.1 * i, 1) for i in range(11)]
    ex = {"terrain": dict(measure_heights=True, measured_points_x=pts_x, measured_points_y=pts_y),
          "env": dict(observe_heights=True, num_observations=70 + 187), "domain_rand": dict(randomize_gravity=False)}
    cfg, S, meta, Bc = T.make_sim("train_noise", N, seed=13, extra=ex)
    hs, hscale, vscale = T.rough_field(seed=2)
    H.bind_height_field(S, Bc, hs, hscale, vscale, 0.0, slope_threshold=0.75 if walls else None)
    T.randomize_dr(Bc, 13)
    Bc.enable_contact_signature()
    Bc.env_origins[0].uniform_(4.0, 19.0, generator=torch.Generator().manual_seed(1))
    Bc.env_origins[1].uniform_(4.0, 19.0, generator=torch.Generator().manual_seed(2))
    ix, iy = (Bc.env_origins[0] / hscale).long(), (Bc.env_origins[1] / hscale).long()
    Bc.env_origins[2] = torch.from_numpy(hs.astype(np.float32))[ix, iy] * vscale + 0.05
    orc = pyoracle.Oracle(S, Bc)
    orc.reset_idx()
    A, simA = T.to_gpu(S, Bc, product=True)
    B, simB = T.to_gpu(S, Bc, product=True)
    Tw, simT = T.to_gpu(S, Bc)
    for Bx, sx in ((A, simA), (B, simB), (Tw, simT)):
        T.sync_from(Bc, Bx, sx, orc)
    rng = np.random.default_rng(0)
    ab, at = {}, {}
    nab = nat = 0
    for step in range(steps):
        a = torch.from_numpy((rng.standard_normal((N, 12)) * (1.0 if step % 2 else 0.3)).astype(np.float32)).cuda()
        for sx in (simA, simB, simT):
            sx.step(a)
        torch.cuda.synchronize()
        nab += int((~T.identical_envs(A, B, N, KEYS, ab)).sum())
        nat += int((~T.identical_envs(A, Tw, N, KEYS, at)).sum())
        align(B, A)
        align(Tw, A)
    print(f"  walls={walls}: product vs product {nab} of {N * steps} env-steps differ {ab}; product vs _sig twin {nat} differ {at}", flush=True)


def plane_digest(N=4096, steps=30):
    """sha256 over the plane product instance's outputs after a free run: equal across two libraries = the same arithmetic on the plane"""
    import hashlib
    cfg, S, meta, Bc, orc = T.gpu_pair("train_noise", N, seed=11)
    A, simA = T.to_gpu(S, Bc, product=True)
    T.sync_from(Bc, A, simA, orc)
    rng = np.random.default_rng(0)
    for step in range(steps):
        simA.step(torch.from_numpy((rng.standard_normal((N, 12)) * (1.0 if step % 2 else 0.3)).astype(np.float32)).cuda())
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for k in KEYS + ["obs_history", "episode_sums"]:
        if A.tensors.get(k) is not None:
            h.update(A.tensors[k].cpu().numpy().tobytes())
    print(f"  plane product instance, {N} envs x {steps} free-running steps: digest {h.hexdigest()[:16]}", flush=True)


if __name__ == "__main__":
    for lib in [None] + sys.argv[1:]:
        if lib:
            H.LIB_PATH, H._lib = os.path.abspath(lib), None
        print(os.path.basename(H.LIB_PATH) if not lib else lib, flush=True)
        for walls in (False, True):
            probe(walls)
        plane_digest()
