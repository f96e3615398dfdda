"""GPU debug: physics substep (aux kernel) and full step vs the oracle on the 'standing' scenario under config variations."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(R, "walk-these-ways_amd", "shims"), os.path.join(R, "walk-these-ways_amd"), os.path.join(R, "oracle"), os.path.join(R, "tests"), R):
    sys.path.insert(0, p)
import numpy as np, torch
import go1sim_host as H, pyoracle
from util import make_sim, randomize_dr, standing_state
N = 64
def run(tag, mode, z=0.28, **over):
    cfg, S, meta, Bc = make_sim("train", N, seed=3, extra={"domain_rand": dict(randomize_gravity=False)})
    for k, v in over.items(): setattr(S, k, v)
    randomize_dr(Bc, 3)
    orc = pyoracle.Oracle(S, Bc); orc.reset_idx()
    standing_state(S, Bc, z=z)
    g = torch.Generator().manual_seed(1)
    Bc.torques.uniform_(-20, 20, generator=g)
    Bg = Bc.clone_to("cuda:0"); sim = H.Go1Sim(S, Bg, 0)
    if mode == "sub":
        orc.physics_substep(); sim.physics_substep()
    else:
        a = np.zeros((N, 12), np.float32)
        orc.step(a); sim.step(torch.from_numpy(a).cuda())
    torch.cuda.synchronize()
    out = []
    for k in ("dof_vel", "contact_forces"):
        d = (Bg.tensors[k].cpu().double() - Bc.tensors[k].double()).abs()
        out.append(f"{k} max {float(d.max()):.4g} bad envs {int((d.reshape(-1, N).max(0).values > 0.05).sum())}")
    print(f"{tag:40s}", " | ".join(out), "faults", Bg.fault_counts.cpu().tolist()[:12])
run("sub z=0.28", "sub")
run("sub z=0.32 (feet only)", "sub", z=0.32)
run("sub z=0.28 no self", "sub", self_collision=0)
run("sub z=0.28 no warm", "sub", warm_start=0)
run("sub z=0.28 1 sweep", "sub", solver_iterations=1)
run("sub z=0.28 0 sweeps", "sub", solver_iterations=0)
run("step z=0.28", "step")
run("step z=0.32", "step", z=0.32)
run("step z=0.28 P control", "step", control_type=0)
