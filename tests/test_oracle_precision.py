"""The oracle in fp32 (oracle/_build/libgo1oracle32.so: the same C source with real = float, Makefile) against the fp64 build.

What separates the two is round-off only.  The parity tests (tests/test_gpu_parity.py, tests/test_emu_parity.py) allow an
environment outside the per-quantity tolerances only if its contact set differs or if THIS fp32 build leaves the fp64 result
there as well; here the fp32 build is measured on its own: on the train.py configuration under N(0,1) actions (robots
falling and tangling: the contact-heavy regime) it leaves the tolerances in far fewer than 1e-3 of the environment-steps, by a
bounded factor — so the 1-2 % outlier budgets of round 2 were not precision (they came from the 8-contact cap's dropped
points), and none is granted now."""
import numpy as np
import torch

from util import make_sim, randomize_dr

TOL = (("root_states", 1e-3, 1e-3), ("dof_pos", 1e-3, 0), ("dof_vel", 2e-2, 1e-3), ("rew_buf", 2e-4, 1e-3), ("torques", 5e-3, 1e-3),
       ("contact_forces", 0.5, 1e-2))


def test_fp32_build_of_the_oracle_stays_within_the_parity_tolerances(oracle_lib):
    N, steps = 1024, 30
    cfg, S, meta, B64 = make_sim("train_noise", N, seed=11)
    randomize_dr(B64, 11)
    B64.enable_contact_signature()
    o64 = oracle_lib.Oracle(S, B64)
    o64.reset_idx()
    B64.episode_length_buf[:] = torch.randint(0, S.max_episode_length, (N,), dtype=torch.int32, generator=torch.Generator().manual_seed(2))
    B32 = B64.clone_to("cpu")
    o32 = oracle_lib.Oracle(S, B32, fp32=True)
    rng = np.random.default_rng(0)
    bad_total, worst, flips = 0, 0.0, 0
    for step in range(steps):
        a = (rng.standard_normal((N, 12)) * (1.0 if step % 2 else 0.3)).astype(np.float32)
        o64.step(a)
        o32.step(a)
        ratio = torch.zeros(N, dtype=torch.float64)
        for k, tol, rt in TOL:
            d = (B32.tensors[k].double() - B64.tensors[k].double()).abs() / (tol + rt * B64.tensors[k].double().abs())
            ratio = torch.maximum(ratio, d.reshape(-1, N).max(0).values)
        bad = ratio > 1.0
        bad_total += int(bad.sum())
        worst = max(worst, float(ratio.max()))
        flips += int((B32.contact_signature != B64.contact_signature).any(0).sum())
        for k, t in B64.tensors.items():           # re-synchronise: one step is compared at a time
            if t is not None and B32.tensors.get(k) is not None:
                B32.tensors[k].copy_(t)
        o32.ctr.common_step_counter, o32.ctr.lag_head, o32.ctr.history_slot = o64.ctr.common_step_counter, o64.ctr.lag_head, o64.ctr.history_slot
    rate = bad_total / (N * steps)
    print(f"fp32 oracle vs fp64 oracle: {N * steps} env-steps, {bad_total} outside the tolerances (rate {rate:.1e}), worst x{worst:.1f}, "
          f"{flips} contact-set flips")
    assert float(B64.contact_forces.abs().max()) > 50.0 and int(B64.reset_buf.sum()) >= 0
    assert rate <= 1e-3 and worst <= 50.0, (rate, worst)
