"""In which ORDER may the contacts of a substep be swept?  The sweeps are the largest single piece of the step kernel's serial spine and their length
is the contact count of the slowest environment of a wavefront; with the four legs' contacts solved SIDE BY SIDE (Gauss-Seidel inside a leg,
block Jacobi over legs) it is the cooperative (trunk, body-body) count + the largest count on one leg.  CPU study on the fp64 oracle
(`go1_oracle_set_solver_order`):
    0  every contact in list order (the contract of rounds 1-4)
    1  all terrain contacts of a leg side by side (round 4's study build)
    2  only the lower-leg contacts (foot, calf) side by side, hip / thigh in list order
    3  THE CONTRACT since round 5: all of a leg's contacts side by side, hip and thigh rows with MASS SPLITTING (the base answers such a row
       n times as strongly in the leg phase, n = legs holding such rows; the true response enters when the legs meet)
    4  every row split
Regimes: (i) limp robots at rest on hips / thighs / trunk after a fall — what separates the orders: plain block Jacobi over hip / thigh rows
(1) does not settle, more sweeps do not cure it; (ii) states from rollouts under N(0, 0.1) / N(0, 0.5) / N(0, 1) actions: one substep solved
with 64 list-order sweeps (the converged reference) and with 2 / 4 / 8 sweeps in every order, distance of the generalised velocity from the
converged one in units of the parity suite's tolerances (base 2e-3 m/s, 1e-2 rad/s; joints 2e-2 rad/s); (iii) closed loop.
Hardware cost of the orders (tools/probes/step_variant_ab.py, gpurun calls r5a-r5f): 2 is no faster than 0 (every wavefront holds a fallen
robot whose cooperative turns the other fifteen environments wait for), 1 and 3 are 8-14 % faster per env.step.

    python tools/solver_order_study.py > profiles/r05_solver_order_study.txt       (round 4's record: profiles/r04_solver_order_study.txt)
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
ORDER_NAMES = {0: 'list order (rounds 1-4)', 1: 'all leg contacts side by side', 2: 'feet + calves side by side', 3: 'all side by side, hip/thigh split', 4: 'all side by side, all rows split'}
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
        L.go1_oracle_set_solver_order(3)
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
        for order in (0, 1, 2, 3, 4):
            for sweeps in (2, 4, 8):
                restore(B, snap)
                L.go1_oracle_set_solver_order(order)
                S.solver_iterations = sweeps
                orc.physics_substep()
                e = np.abs(velocity(B) - ref) / TOL
                errs.setdefault((order, sweeps), []).append(e.max(0))
        L.go1_oracle_set_solver_order(3)
        S.solver_iterations = 4
        restore(B, snap)
    counts, legmax = np.concatenate(counts), np.concatenate(legmax)
    print(f"\n{name}: {len(counts)} (environment, substep) samples; bodies in contact per environment: mean {counts.mean():.1f}, max {counts.max()};"
          f" largest count on ONE leg: mean {legmax.mean():.1f}, max {legmax.max()}")
    print("  worst component of |v - v_converged| / tolerance:      median       90 %       99 %        max    share > 1")
    for (order, sweeps), e in sorted(errs.items()):
        e = np.concatenate(e)
        q = np.quantile(e, [0.5, 0.9, 0.99])
        print(f"  {ORDER_NAMES[order]:34s} {sweeps:2d} sweeps        {q[0]:10.3f} {q[1]:10.3f} {q[2]:10.3f} {e.max():10.2f} {100 * (e > 1).mean():9.1f} %")
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
    orc.L.go1_oracle_set_solver_order(3)
    finite = bool(torch.isfinite(B.root_states).all() and torch.isfinite(B.dof_vel).all())
    return resets, zsum / steps, wmax, finite


def fallen_at_rest(order, sweeps, seed=3, n=64):
    """limp robots dropped in random orientations; after 3 s: the largest base |v| and |omega| over 40 substeps, per environment (a robot at rest
    on hips / thighs / trunk is what the rollouts above never hold: the training configuration ends an episode soon after a fall)"""
    cfg, S, meta, B = make_sim("train", n, extra={"domain_rand": dict(randomize_gravity=False)})
    g = torch.Generator().manual_seed(seed)
    standing_state(S, B, 0.30)
    q = torch.randn(4, n, generator=g)
    B.root_states[3:7] = q / q.norm(dim=0, keepdim=True)
    lo = torch.tensor([-0.86, -0.68, -2.81] * 4).unsqueeze(1)
    hi = torch.tensor([0.86, 4.50, -0.89] * 4).unsqueeze(1)
    B.dof_pos[:] = lo + (hi - lo) * torch.rand(12, n, generator=g)
    S.solver_iterations = sweeps
    orc = pyoracle.Oracle(S, B)
    orc.L.go1_oracle_set_solver_order(order)
    B.torques.zero_()
    for _ in range(600):
        orc.physics_substep()
    wmax, vmax = torch.zeros(n), torch.zeros(n)
    for _ in range(40):
        orc.physics_substep()
        wmax = torch.maximum(wmax, B.root_states[10:13].norm(dim=0))
        vmax = torch.maximum(vmax, B.root_states[7:10].norm(dim=0))
    orc.L.go1_oracle_set_solver_order(3)
    ncont = (B.contact_forces.view(17, 3, -1).norm(dim=1) > 0).sum(0).float()
    return wmax.numpy(), vmax.numpy(), float(ncont.mean())


def main():
    print(__doc__.rsplit("\n\n", 1)[0])
    print("\nLimp robots dropped in random orientations and joint angles, 64 environments, 3 s of settling, then 40 substeps: base creep at rest")
    print("  order                              sweeps   bodies in contact   |v| median / 90 % / max [m/s]      |omega| median / 90 % / max [rad/s]")
    for order in (0, 1, 2, 3, 4):
        for sweeps in (4, 8):
            w, v, nc = fallen_at_rest(order, sweeps)
            print(f"  {ORDER_NAMES[order]:34s} {sweeps:4d} {nc:12.1f}          {np.median(v):.4f} / {np.quantile(v, 0.9):.4f} / {v.max():.4f}"
                  f"            {np.median(w):.4f} / {np.quantile(w, 0.9):.4f} / {w.max():.4f}")
    for name, std, steps, every in (("standing / shuffling (actions N(0, 0.1))", 0.1, 60, 6), ("walking-scale actions N(0, 0.5)", 0.5, 60, 6),
                                    ("falling and tangling (actions N(0, 1): the parity suite's regime)", 1.0, 90, 6)):
        cfg, S, meta, B = make_sim("train", N, extra={"domain_rand": dict(randomize_gravity=False)})
        standing_state(S, B, 0.32)
        orc = pyoracle.Oracle(S, B)
        study(name, std, steps, every, S, B, orc, np.random.default_rng(11))
    print("\nClosed loop, 300 policy steps (1200 substeps) of N(0, 0.5) actions from standing, same action stream; episodes ended (falls + time-outs),")
    print("mean base height, largest base angular velocity, all state finite:")
    for order, sweeps in ((0, 4), (1, 4), (2, 4), (3, 4), (4, 4)):
        r, z, w, fin = closed_loop(order, sweeps, 0.5, 300, 5)
        print(f"  {ORDER_NAMES[order]:34s} {sweeps} sweeps: {r:4d} episodes ended, mean height {z:.3f} m, max |omega| {w:6.2f} rad/s, finite {fin}")


if __name__ == "__main__":
    main()
