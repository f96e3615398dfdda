"""Replay a captured fault (tools/fault_hunt.py -> fault_XXXX.npz): re-run exactly that policy step, substep by substep,
in the fp64 oracle (CPU, default) and/or in the HIP kernel (--hip, GPU box), and print the trace.

    python tools/fault_replay.py gpurun_out/faults/fault_0000.npz [--hip] [--env 0]
"""
import argparse
import ctypes
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "walk-these-ways_amd")
for p in (os.path.join(PKG, "shims"), PKG, os.path.join(REPO, "oracle"), REPO):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402


def load_case(path, which, copies, device="cpu"):
    import go1sim_abi as abi
    import go1sim_host as H
    from go1_gym.envs.base.legged_robot_config import make_cfg
    from scripts.train_config import apply_train_config
    d = np.load(path, allow_pickle=False)
    S = abi.Go1SimConfig()
    raw = d["config"].tobytes()
    assert len(raw) == ctypes.sizeof(S), "fixture written by a different ABI"
    ctypes.memmove(ctypes.addressof(S), raw, len(raw))
    N0 = int(d["num_envs"])
    S.env_id_offset = int(S.env_id_offset) + int(d["env_ids"][which])
    S.num_envs = copies
    _, meta = H.build_sim_config(apply_train_config(make_cfg(), num_envs=copies))
    B = H.SimBuffers(S, meta, "cpu")
    for k, t in B.tensors.items():
        key = "pre_" + k
        if t is None or key not in d.files:
            continue
        a = torch.from_numpy(np.ascontiguousarray(d[key]))
        if t.dim() >= 1 and t.shape[-1] == copies and a.shape[-1] == len(d["env_ids"]) and a.dim() == 2:
            t.reshape(-1, copies).copy_(a[:, which:which + 1].expand(-1, copies))
        elif t.dim() >= 1 and t.shape[0] == copies and a.shape[0] == len(d["env_ids"]):
            t.copy_(a[which:which + 1].expand_as(t))
        elif t.shape == a.shape:
            t.copy_(a)
    B.fault_flags.zero_(); B.fault_counts.zero_()
    act = np.repeat(d["actions"][which:which + 1], copies, axis=0).astype(np.float32)
    return d, S, B if device == "cpu" else B.clone_to(device), act, N0


def show(tag, B, e=0):
    r = B.root_states[:, e].double().cpu().numpy()
    cf = B.contact_forces.view(17, 3, -1)[:, :, e].double().cpu().numpy()
    print(f"  {tag}: pos {np.round(r[:3], 4)} quat {np.round(r[3:7], 4)} v {np.round(r[7:10], 3)} w {np.round(r[10:13], 3)}")
    print(f"      q  {np.round(B.dof_pos[:, e].double().cpu().numpy(), 3)}")
    print(f"      qd {np.round(B.dof_vel[:, e].double().cpu().numpy(), 2)}")
    print(f"      tau {np.round(B.torques[:, e].double().cpu().numpy(), 2)}")
    nz = [(b, np.round(cf[b], 1)) for b in range(17) if np.abs(cf[b]).max() > 0 or not np.isfinite(cf[b]).all()]
    print(f"      contact forces {nz}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("npz")
    ap.add_argument("--env", type=int, default=0, help="index into the captured env list")
    ap.add_argument("--hip", action="store_true")
    args = ap.parse_args()
    import go1sim_host as H
    import pyoracle
    d, S, B, act, N0 = load_case(args.npz, args.env, 1)
    words = int(d["fault_words"][args.env])
    print(f"env {int(d['env_ids'][args.env])} of {N0}, counter {int(d['counter'])}, lag_head {int(d['lag_head'])}, fault "
          f"{[H.FAULT_NAMES[b] for b in H.FAULT_NAMES if words >> b & 1]}")
    print(f"  mu {float(B.friction_coeffs[0]):.3f} rest {float(B.restitutions[0]):.3f} payload {float(B.payloads[0]):.3f} "
          f"com {B.com_displacements[:, 0].tolist()} ep_len {int(B.episode_length_buf[0])} action {np.round(act[0], 2)}")
    show("pre  ", B)
    # oracle, substep by substep: the piecewise entry points are sub-ranges of the step
    orc = pyoracle.Oracle(S, B)
    orc.ctr.common_step_counter = int(d["counter"]); orc.ctr.lag_head = int(d["lag_head"])
    a_clip = np.clip(act, -S.clip_actions, S.clip_actions)
    for sub in range(S.decimation):
        orc.compute_torques(np.ascontiguousarray(a_clip.T))
        orc.physics_substep()
        show(f"oracle sub {sub}", B)
    if args.hip:
        d, S, Bg, act, _ = load_case(args.npz, args.env, 16, "cuda:0")
        sim = H.Go1Sim(S, Bg, 0)
        sim.set_counters(int(d["counter"]), int(d["lag_head"]))
        a_soa = torch.from_numpy(np.ascontiguousarray(np.clip(act, -S.clip_actions, S.clip_actions).T)).cuda()
        for sub in range(S.decimation):
            sim.compute_torques(a_soa)
            sim.physics_substep()
            torch.cuda.synchronize()
            show(f"hip    sub {sub}", Bg)
            w = int(Bg.fault_flags[0])
            print(f"      fault word {[H.FAULT_NAMES[b] for b in H.FAULT_NAMES if w >> b & 1]}")
        d, S, Bg, act, _ = load_case(args.npz, args.env, 16, "cuda:0")
        sim = H.Go1Sim(S, Bg, 0)
        sim.set_counters(int(d["counter"]), int(d["lag_head"]))
        sim.step(torch.from_numpy(act).cuda())
        torch.cuda.synchronize()
        w = int(Bg.fault_flags[0])
        print(f"  full HIP step: fault word {[H.FAULT_NAMES[b] for b in H.FAULT_NAMES if w >> b & 1]} reward {float(Bg.rew_buf[0])} reset {int(Bg.reset_buf[0])}")


if __name__ == "__main__":
    main()
