"""GPU debug: one environment-step of tests/test_gpu_parity.py::test_product_instances_match_oracle[hf] replayed on the hardware with
several builds of libgo1sim.so (compiler-flag / code-path variants in csrc/variants/), against the fp64 and fp32 oracle.
    python tools/debug/hf_env_replay.py ENV STEP lib1.so lib2.so ...        (the oracle alone regenerates the state: deterministic)"""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(R, "walk-these-ways_amd", "shims"), os.path.join(R, "walk-these-ways_amd"), os.path.join(R, "oracle"), os.path.join(R, "tests"), R):
    sys.path.insert(0, p)
import numpy as np, torch
import go1sim_host as H, pyoracle
from util import make_sim, randomize_dr
from test_gpu_parity import rough_field
N, E, STEP = 4096, int(sys.argv[1]), int(sys.argv[2])
libs = sys.argv[3:] or [H.LIB_PATH]
pts_x = [round(-0.8 + 0.1 * i, 1) for i in range(17)]; pts_y = [round(-0.5 + 0.1 * i, 1) for i in range(11)]
ex = {"terrain": dict(measure_heights=True, measured_points_x=pts_x, measured_points_y=pts_y),
      "env": dict(observe_heights=True, num_observations=70 + 187), "domain_rand": dict(randomize_gravity=False)}
def build(n):
    cfg, S, meta, B = make_sim("train_noise", n, seed=13, extra=ex)
    hs, hscale, vscale = rough_field(seed=2)
    H.bind_height_field(S, B, hs, hscale, vscale, 0.0, slope_threshold=None)
    return S, B, hs, hscale, vscale
S, Bc, hs, hscale, vscale = build(N)
randomize_dr(Bc, 13)
Bc.env_origins[0].uniform_(4.0, 19.0, generator=torch.Generator().manual_seed(1))
Bc.env_origins[1].uniform_(4.0, 19.0, generator=torch.Generator().manual_seed(2))
ix = (Bc.env_origins[0] / hscale).long(); iy = (Bc.env_origins[1] / hscale).long()
Bc.env_origins[2] = torch.from_numpy(hs.astype(np.float32))[ix, iy] * vscale + 0.05
orc = pyoracle.Oracle(S, Bc); orc.reset_idx()
S1, B1, *_ = build(16)
def copy_env(src, dst, e):
    for k, t in src.tensors.items():
        d = dst.tensors.get(k)
        if t is None or d is None: continue
        if t.dim() >= 1 and t.shape[-1] == N and d.shape[-1] == 16: d[..., :] = t[..., e:e + 1]
        elif t.dim() >= 1 and t.shape[0] == N and d.shape[0] == 16: d[:] = t[e:e + 1]
rng = np.random.default_rng(0)
for step in range(STEP + 1):
    a = (rng.standard_normal((N, 12)) * (1.0 if step % 2 else 0.3)).astype(np.float32)
    if step == STEP:
        copy_env(Bc, B1, E)
        a1 = np.repeat(a[E:E + 1], 16, 0)
        f = lambda t: [round(float(x), 5) for x in t]
        res = {}
        for tag, fp32 in (("o64", False), ("o32", True)):
            B = B1.clone_to("cpu"); o = pyoracle.Oracle(S1, B, fp32=fp32)
            o.ctr.common_step_counter, o.ctr.lag_head, o.ctr.history_slot = orc.ctr.common_step_counter, orc.ctr.lag_head, orc.ctr.history_slot
            o.step(a1); res[tag] = B
        print("o32 - o64 qd", f(res["o32"].dof_vel[:, 0] - res["o64"].dof_vel[:, 0]))
        for path in libs:
            lib = H.bind_library(ctypes.CDLL(os.path.abspath(path)))
            Bg = B1.clone_to("cuda:0"); sim = H.Go1Sim(S1, Bg, 0, lib=lib)
            sim.set_counters(orc.ctr.common_step_counter, orc.ctr.lag_head)
            sim.step(torch.from_numpy(a1).cuda()); torch.cuda.synchronize()
            dr = Bg.root_states[:, 0].cpu() - res["o64"].root_states[:, 0]
            print("   root kernel - o64", f(dr), " o32 - o64", f(res["o32"].root_states[:, 0] - res["o64"].root_states[:, 0]))
            d = Bg.dof_vel[:, 0].cpu() - res["o64"].dof_vel[:, 0]
            # the same environment alone (a partial wavefront: plain-FMA torque model instead of the MFMA one, helpers mostly idle)
            S0, B0, *_ = build(1)
            for k, t in B1.tensors.items():
                d0 = B0.tensors.get(k)
                if t is None or d0 is None: continue
                if t.dim() >= 1 and t.shape[-1] == 16 and d0.shape[-1] == 1: d0[..., :] = t[..., :1]
                elif t.dim() >= 1 and t.shape[0] == 16 and d0.shape[0] == 1: d0[:] = t[:1]
            Bg0 = B0.clone_to("cuda:0"); sim0 = H.Go1Sim(S0, Bg0, 0, lib=lib)
            sim0.set_counters(orc.ctr.common_step_counter, orc.ctr.lag_head)
            sim0.step(torch.from_numpy(a1[:1]).cuda()); torch.cuda.synchronize()
            print("   ALONE (N = 1): root kernel - o64", f(Bg0.root_states[:, 0].cpu() - res["o64"].root_states[:, 0]), " qd max",
                  round(float((Bg0.dof_vel[:, 0].cpu() - res["o64"].dof_vel[:, 0]).abs().max()), 5))
            print(f"{os.path.basename(path):28s} kernel - o64 qd {f(d)}  max {float(d.abs().max()):.5f}  torque diff {float((Bg.torques[:,0].cpu()-res['o64'].torques[:,0]).abs().max()):.2e}"
                  f"  slots agree {bool((Bg.dof_vel.cpu() - Bg.dof_vel[:, :1].cpu()).abs().max() == 0)}")
    orc.step(a)
