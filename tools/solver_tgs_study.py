"""A TGS-like solve beside the contract's PGS, in the regimes where the 4-sweep PGS residual shows (profiles/r03_solver_convergence.txt): the
reference runs PhysX with solver_type 1 (TGS), 4 position iterations, 0 velocity iterations (legged_robot_config.py:410-414); this repo's
contract is a 4-sweep PGS.  TGS re-evaluates the constraint errors between its iterations; `go1_oracle_set_tgs_like` (oracle/go1_oracle.c) restates
that for the velocity-level solve with frozen Jacobians: mode 1 = the targets follow the error that is left, the pose is integrated with the final
velocity; mode 2 = the pose also advances by h / N with every sweep's velocity (the TGS stepping).  STUDY ONLY (fp64 oracle, CPU): the kernel runs
the contract; what a TGS-like sweep would cost there is one fused multiply-add per contact and sweep (the accumulated normal displacement) and, in
mode 2, 18 more per environment and sweep (the pose accumulator) — noise beside the 3 x 18-wide row products of a contact's update.

    python tools/solver_tgs_study.py > profiles/r05_tgs_like_study.txt
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(REPO, "tests"), os.path.join(REPO, "walk-these-ways_amd", "shims"), os.path.join(REPO, "walk-these-ways_amd"),
          os.path.join(REPO, "oracle"), REPO):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import pyoracle  # noqa: E402
from util import make_sim, standing_state  # noqa: E402

POSES = ["limp, dropped on its feet (collapses onto the belly and the folded legs)", "limp on the belly, legs folded (trunk corners + calves + thighs)",
         "limp on its side, tangled (terrain points, leg-leg contacts, limit rows)", "limp on its back"]
ARMS = [("PGS (the contract)", 0), ("TGS-like, targets only", 1), ("TGS-like, targets + stepping", 2)]
MG = 11.309932 * 9.8


def poses(S, B):
    standing_state(S, B, 0.30)
    B.root_states[2, 1] = 0.10; B.dof_pos[:, 1] = torch.tensor([0.0, 1.3, -2.6] * 4)
    B.root_states[2, 2] = 0.12; B.root_states[3, 2] = np.sin(np.pi / 4); B.root_states[6, 2] = np.cos(np.pi / 4)
    B.root_states[2, 3] = 0.15; B.root_states[3, 3] = 1.0; B.root_states[6, 3] = 0.0


def rest(sweeps, mode, settle=800, window=40):
    """limp robots settle for 4 s; residual motion, support force and sinking over the 40 substeps after that"""
    cfg, S, meta, B = make_sim("train", 4, extra={"domain_rand": dict(randomize_gravity=False)})
    S.solver_iterations = sweeps
    poses(S, B)
    orc = pyoracle.Oracle(S, B)
    orc.L.go1_oracle_set_tgs_like(mode)
    try:
        B.torques.zero_()
        for it in range(settle):
            orc.physics_substep()
        w, v, fz = np.zeros(4), np.zeros(4), np.zeros(4)
        z0 = B.root_states[2].clone()
        for it in range(window):
            orc.physics_substep()
            w = np.maximum(w, B.root_states[10:13].norm(dim=0).numpy())
            v = np.maximum(v, B.root_states[7:10].norm(dim=0).numpy())
            fz += B.contact_forces.view(17, 3, 4)[:, 2].sum(0).numpy() / window
        sink = (z0 - B.root_states[2]).numpy() / (window * 0.005)
    finally:
        orc.L.go1_oracle_set_tgs_like(0)
    return w, v, np.abs(fz - MG) / MG, sink


def impact(sweeps, mode, N=32, steps=80):
    """limp robots with folded legs dropped flat on the belly (trunk box: half height 0.057 m) at -1 ... -4 m/s: lowest base height reached
    (below 0.057 = the trunk corners under the ground), the highest upward base speed afterwards (rebound; restitution is 0) and the base
    height after 0.4 s — what re-evaluated errors change in the regime where a 4-sweep solve is soft: an impact on many points at once"""
    cfg, S, meta, B = make_sim("train", N, extra={"domain_rand": dict(randomize_gravity=False)})
    S.solver_iterations = sweeps
    standing_state(S, B, 0.30)
    B.dof_pos[:] = torch.tensor([0.0, 1.3, -2.6] * 4)[:, None]
    B.root_states[2] = 0.20
    B.root_states[9] = -torch.linspace(1.0, 4.0, N)
    orc = pyoracle.Oracle(S, B)
    orc.L.go1_oracle_set_tgs_like(mode)
    try:
        B.torques.zero_()
        zmin = np.full(N, 1e9)
        vup = np.zeros(N)
        for it in range(steps):
            orc.physics_substep()
            zmin = np.minimum(zmin, B.root_states[2].numpy())
            vup = np.maximum(vup, B.root_states[9].numpy())
        zend = B.root_states[2].numpy().copy()
    finally:
        orc.L.go1_oracle_set_tgs_like(0)
    return zmin, vup, zend


def main():
    print(__doc__.split("\n\n")[0].replace("\n", " "))
    print()
    print("A. Limp robots (zero joint torques) settle for 4 s (800 substeps); residuals over the 40 substeps after that.")
    print("   columns: max |omega_base| rad/s, max |v_base| m/s, |sum F_z - m g| / m g (time mean), sinking speed of the base m/s")
    res = {(name, sweeps): rest(sweeps, mode) for name, mode in ARMS for sweeps in (4, 8)}
    ref = rest(64, 0)
    for p, pose in enumerate(POSES):
        print(f"\n{pose}")
        for (name, sweeps), (w, v, fe, sink) in res.items():
            print(f"  {name:30s} {sweeps:2d} sweeps: omega {w[p]:8.5f}  v {v[p]:8.5f}  force error {fe[p]:8.5f}  sinking {sink[p]:9.6f}")
        w, v, fe, sink = ref
        print(f"  {'PGS, converged':30s} 64 sweeps: omega {w[p]:8.5f}  v {v[p]:8.5f}  force error {fe[p]:8.5f}  sinking {sink[p]:9.6f}")
    print()
    print("B. 32 limp robots with folded legs dropped flat on the belly from 0.20 m at -1 ... -4 m/s, 80 substeps (0.4 s): lowest base height reached,")
    print("   highest upward base speed afterwards (rebound), base height at the end; median | worst over the robots")
    for name, mode in ARMS:
        for sweeps in (4, 8):
            zm, vu, ze = impact(sweeps, mode)
            print(f"  {name:30s} {sweeps:2d} sweeps: lowest base {np.median(zm):7.4f} | {zm.min():7.4f} m   rebound {np.median(vu):6.3f} | {vu.max():6.3f} m/s   at rest {np.median(ze):7.4f} m")
    zm, vu, ze = impact(64, 0)
    print(f"  {'PGS, converged':30s} 64 sweeps: lowest base {np.median(zm):7.4f} | {zm.min():7.4f} m   rebound {np.median(vu):6.3f} | {vu.max():6.3f} m/s   at rest {np.median(ze):7.4f} m")


if __name__ == "__main__":
    main()
