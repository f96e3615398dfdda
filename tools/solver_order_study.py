"""Would a lane-per-leg kernel that solves the four legs' terrain contacts SIDE BY SIDE (Gauss-Seidel inside a leg, the legs' velocity changes
added up: block Jacobi over legs) still converge like the contract's sweep over the contact list?  The sweeps are 32 % of the step kernel's serial
spine and their length is the contact COUNT of the slowest environment of a wavefront; side by side it would be the largest count of one LEG.
CPU study on the fp64 oracle (`go1_oracle_set_solver_order`, a study switch — the contract is unchanged): at states taken from rollouts, one
physics substep is solved with 64 list-order sweeps (the converged reference of the same contact model) and with 2 / 4 / 8 sweeps in both orders;
reported is the distance of the resulting generalised velocity from the converged one, in units of the parity suite's tolerances
(base 2e-3 m/s, 1e-2 rad/s; joints 2e-2 rad/s).

    python tools/solver_order_study.py > profiles/r04_solver_order_study.txt
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

N = 64
TOL = np.concatenate([np.full(3, 2e-3), np.full(3, 1e-2), np.full(12, 2e-2)])[:, None]      # v_lin, v_ang, joint rates


def snapshot(B, orc):
    return ({k: v.clone() for k, v in B.tensors.items() if v is not None},)


def restore(B, snap):
    for k, v in snap[0].items():
        B.tensors[k].copy_(v)


def velocity(B):
    return torch.cat([B.root_states[7:13], B.dof_vel]).numpy().astype(np.float64).copy()


def contacts_per_leg(B):
    f = B.contact_forces.view(17, 3, -1).norm(dim=1) > 0                # (17, N) bodies with a contact force
    per_leg = torch.stack([f[1 + 4 * leg:5 + 4 * leg].sum(0) for leg in range(4)])
    return f.sum(0).numpy(), per_leg.max(0).values.numpy()


def study(name, action_std, steps, every, S, B, orc, rng):
    L = orc.L
    errs = {}
    counts, legmax = [], []
    for t in range(steps):
        a = (rng.standard_normal((N, 12)) * action_std).astype(np.float32)
        L.go1_oracle_set_solver_order(0)
        S.solver_iterations = 4
        orc.step(a)
        if t % every:
            continue
        snap = snapshot(B, orc)
        S.solver_iterations = 64
        orc.physics_substep()
        ref = velocity(B)
        c, lm = contacts_per_leg(B)
        counts.append(c); legmax.append(lm)
        for order in (0, 1):
            for sweeps in (2, 4, 8):
                restore(B, snap)
                L.go1_oracle_set_solver_order(order)
                S.solver_iterations = sweeps
                orc.physics_substep()
                e = np.abs(velocity(B) - ref) / TOL
                errs.setdefault((order, sweeps), []).append(e.max(0))
        L.go1_oracle_set_solver_order(0)
        S.solver_iterations = 4
        restore(B, snap)
    counts, legmax = np.concatenate(counts), np.concatenate(legmax)
    print(f"\n{name}: {len(counts)} (environment, substep) samples; bodies in contact per environment: mean {counts.mean():.1f}, max {counts.max()};"
          f" largest count on ONE leg: mean {legmax.mean():.1f}, max {legmax.max()}")
    print("  worst component of |v - v_converged| / tolerance:      median       90 %       99 %        max    share > 1")
    for (order, sweeps), e in sorted(errs.items()):
        e = np.concatenate(e)
        q = np.quantile(e, [0.5, 0.9, 0.99])
        print(f"  {'legs side by side' if order else 'list order (contract)':22s} {sweeps:2d} sweeps        {q[0]:10.3f} {q[1]:10.3f} {q[2]:10.3f} {e.max():10.2f} {100 * (e > 1).mean():9.1f} %")
    return errs


def closed_loop(order, sweeps, action_std, steps, seed):
    cfg, S, meta, B = make_sim("train", N, extra={"domain_rand": dict(randomize_gravity=False)})
    standing_state(S, B, 0.32)
    orc = pyoracle.Oracle(S, B)
    orc.L.go1_oracle_set_solver_order(order)
    S.solver_iterations = sweeps
    rng = np.random.default_rng(seed)
    resets, zsum, wmax = 0, 0.0, 0.0
    for t in range(steps):
        orc.step((rng.standard_normal((N, 12)) * action_std).astype(np.float32))
        resets += int(B.reset_buf.sum())
        zsum += float(B.root_states[2].mean())
        wmax = max(wmax, float(B.root_states[10:13].norm(dim=0).max()))
    orc.L.go1_oracle_set_solver_order(0)
    finite = bool(torch.isfinite(B.root_states).all() and torch.isfinite(B.dof_vel).all())
    return resets, zsum / steps, wmax, finite


def main():
    print(__doc__.split("\n\n")[0])
    for name, std, steps, every in (("standing / shuffling (actions N(0, 0.1))", 0.1, 60, 6), ("walking-scale actions N(0, 0.5)", 0.5, 60, 6),
                                    ("falling and tangling (actions N(0, 1): the parity suite's regime)", 1.0, 90, 6)):
        cfg, S, meta, B = make_sim("train", N, extra={"domain_rand": dict(randomize_gravity=False)})
        standing_state(S, B, 0.32)
        orc = pyoracle.Oracle(S, B)
        study(name, std, steps, every, S, B, orc, np.random.default_rng(11))
    print("\nClosed loop, 300 policy steps (1200 substeps) of N(0, 0.5) actions from standing, same action stream; episodes ended (falls + time-outs),")
    print("mean base height, largest base angular velocity, all state finite:")
    for order, sweeps in ((0, 4), (1, 4), (1, 8)):
        r, z, w, fin = closed_loop(order, sweeps, 0.5, 300, 5)
        print(f"  {'legs side by side' if order else 'list order (contract)':22s} {sweeps} sweeps: {r:4d} episodes ended, mean height {z:.3f} m, max |omega| {w:6.2f} rad/s, finite {fin}")


if __name__ == "__main__":
    main()
