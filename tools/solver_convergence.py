"""PGS with 4 sweeps here vs PhysX's TGS with 4 position iterations in the reference (legged_robot_config.py:410-414: solver_type 1,
num_position_iterations 4, num_velocity_iterations 0): the two are NOT the same algorithm — TGS re-evaluates the constraint
errors between its iterations and converges faster per iteration — and the reference's binary cannot be run, so the deviation is
stated and measured instead: residuals of the 4-sweep solve on the resting poses where it matters, against the converged
(64-sweep) solve of the same contact model.  CPU only (the oracle; the kernel reproduces it, tests/test_emu_parity.py):

    python tools/solver_convergence.py > profiles/r03_solver_convergence.txt
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
         "limp on its side, tangled (13 terrain points, 2 leg-leg contacts, limit rows of 2 legs)", "limp on its back"]


def run(sweeps, settle=800, window=40):
    cfg, S, meta, B = make_sim("train", 4, extra={"domain_rand": dict(randomize_gravity=False)})
    S.solver_iterations = sweeps
    standing_state(S, B, 0.30)
    B.root_states[2, 1] = 0.10; B.dof_pos[:, 1] = torch.tensor([0.0, 1.3, -2.6] * 4)
    B.root_states[2, 2] = 0.12; B.root_states[3, 2] = np.sin(np.pi / 4); B.root_states[6, 2] = np.cos(np.pi / 4)
    B.root_states[2, 3] = 0.15; B.root_states[3, 3] = 1.0; B.root_states[6, 3] = 0.0
    orc = pyoracle.Oracle(S, B)
    B.torques.zero_()
    for it in range(settle):
        orc.physics_substep()
    w, v, fz, pen = np.zeros(4), np.zeros(4), np.zeros(4), np.zeros(4)
    z0 = B.root_states[2].clone()
    for it in range(window):
        orc.physics_substep()
        w = np.maximum(w, B.root_states[10:13].norm(dim=0).numpy())
        v = np.maximum(v, B.root_states[7:10].norm(dim=0).numpy())
        fz += B.contact_forces.view(17, 3, 4)[:, 2].sum(0).numpy() / window
    sink = (z0 - B.root_states[2]).numpy() / (window * 0.005)          # residual sinking speed of the base
    return w, v, fz, sink


def main():
    mg = 11.309932 * 9.8
    print(__doc__.split("\n\n")[0].replace("\n", " "))
    print()
    print("Limp robots (zero joint torques) settle for 4 s (800 substeps); residuals over the 40 substeps after that.")
    print("columns: max |omega_base| rad/s, max |v_base| m/s, |sum F_z - m g| / m g (time mean), sinking speed of the base m/s")
    results = {}
    for sweeps in (1, 2, 4, 8, 16, 64):
        results[sweeps] = run(sweeps)
    for p, name in enumerate(POSES):
        print(f"\n{name}")
        for sweeps, (w, v, fz, pen) in results.items():
            tag = "  <- reference setting (4 position iterations)" if sweeps == 4 else ""
            print(f"  {sweeps:3d} sweeps: omega {w[p]:8.5f}  v {v[p]:8.5f}  force error {abs(fz[p] - mg) / mg:8.5f}  sinking {pen[p]:9.6f}{tag}")


if __name__ == "__main__":
    main()
