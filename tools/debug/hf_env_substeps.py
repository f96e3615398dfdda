"""GPU debug: ONE environment-step of the hf product run, substep by substep (torque model + physics substep through the piecewise entry
points), on the hardware kernel, on the SIMT emulation of the same kernel sources (CPU) and on the fp64 / fp32 oracle — each free-running
from the same state.  Locates the substep and the bodies where the hardware leaves the emulation.
    python tools/debug/hf_env_substeps.py ENV STEP"""
import ctypes, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(R, "walk-these-ways_amd", "shims"), os.path.join(R, "walk-these-ways_amd"), os.path.join(R, "oracle"), os.path.join(R, "tests"),
          os.path.join(R, "tests", "emu"), R):
    sys.path.insert(0, p)
import numpy as np, torch
import go1sim_host as H, pyoracle, emu_sim
from util import make_sim, randomize_dr
from test_gpu_parity import rough_field
N, E, STEP = 4096, int(sys.argv[1]), int(sys.argv[2])
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
B1.enable_contact_signature()
def copy_env(src, dst, e):
    for k, t in src.tensors.items():
        d = dst.tensors.get(k)
        if t is None or d is None: continue
        if t.dim() >= 1 and t.shape[-1] == N and d.shape[-1] == 16: d[..., :] = t[..., e:e + 1]
        elif t.dim() >= 1 and t.shape[0] == N and d.shape[0] == 16: d[:] = t[e:e + 1]
rng = np.random.default_rng(0)
f = lambda t: [round(float(x), 5) for x in t]
for step in range(STEP + 1):
    a = (rng.standard_normal((N, 12)) * (1.0 if step % 2 else 0.3)).astype(np.float32)
    if step == STEP:
        copy_env(Bc, B1, E)
        act = np.clip(np.repeat(a[E:E + 1], 16, 0), -S1.clip_actions, S1.clip_actions).T.copy()
        impl = {}
        for tag in ("o64", "o32"):
            B = B1.clone_to("cpu"); o = pyoracle.Oracle(S1, B, fp32=(tag == "o32"))
            o.ctr.common_step_counter, o.ctr.lag_head, o.ctr.history_slot = orc.ctr.common_step_counter, orc.ctr.lag_head, orc.ctr.history_slot
            impl[tag] = (B, o)
        Be = B1.clone_to("cpu"); se = emu_sim.EmuSim(S1, Be); se.set_counters(orc.ctr.common_step_counter, orc.ctr.lag_head); impl["emu"] = (Be, se)
        Bg = B1.clone_to("cuda:0"); sg = H.Go1Sim(S1, Bg, 0); sg.set_counters(orc.ctr.common_step_counter, orc.ctr.lag_head); impl["hw"] = (Bg, sg)
        for sub in range(4):
            for tag, (B, o) in impl.items():
                if tag in ("o64", "o32"):
                    o.compute_torques(act); o.physics_substep()
                elif tag == "emu":
                    o.compute_torques(torch.from_numpy(act)); o.physics_substep()
                else:
                    o.compute_torques(torch.from_numpy(act).cuda()); o.physics_substep(); torch.cuda.synchronize()
            lh = impl["o64"][1].ctr.lag_head
            impl["o32"][1].ctr.lag_head = lh
            for tag in ("emu", "hw"):
                impl[tag][1].set_counters(orc.ctr.common_step_counter, lh)
            g = lambda tag, k: impl[tag][0].tensors[k][..., 0].cpu().double() if impl[tag][0].tensors[k].shape[-1] == 16 else None
            print(f"--- substep {sub}")
            for k in ("dof_vel", "root_states", "torques"):
                print(f"  {k:12s} hw-emu  {f(g('hw', k) - g('emu', k))}")
                print(f"  {k:12s} emu-o64 {f(g('emu', k) - g('o64', k))}")
            cf = {t: impl[t][0].contact_forces.view(17, 3, -1)[:, :, 0].cpu().double() for t in impl}
            nz = (cf["o64"].abs().sum(1) + cf["hw"].abs().sum(1) > 0).nonzero().flatten().tolist()
            for b in nz:
                print(f"  body {b:2d} force hw {f(cf['hw'][b])} emu {f(cf['emu'][b])} o64 {f(cf['o64'][b])}")
            sig = lambda t: [hex(int(x) & 0xffffffff) for x in impl[t][0].contact_signature[:4, 0]]
            print(f"  signature hw {sig('hw')} emu {sig('emu')} o64 {sig('o64')}")
    orc.step(a)
