"""HIP step (through the C-ABI) vs the CPU oracle on identical seeded inputs.  fp32 kernel vs fp64 oracle.

Tolerances are stated per quantity where they are applied (one physics substep from identical state: q, root 2e-4, qd 3e-3,
contact forces 5e-2 N + 2e-3 rel; full step: q, root 1e-3, qd 2e-2, rewards 2e-4, observations 3e-3, torques 5e-3 — on the
height-field relief: root 3e-3, observations 5e-3, torques 2e-2 — each with the relative part given next to it).  There is NO free outlier budget.  An environment may exceed a tolerance only if it is
ATTRIBUTED, in one of three checkable ways (a step attributed to precision alone, rule (b), stays below ATTRIBUTED_BOUND = 50 x the tolerance):
  (a) contact set: the contact points / self pairs / limit-row legs the solver listed in some substep, the height-field cell /
      corner a listed point came from, or the ACTIVE SET the solve ended in (pressing contacts, contacts on the friction cone,
      limit rows carrying an impulse) differ between kernel and oracle — both record them (include/go1sim.h
      `contact_signature`): a point sitting on an activation threshold or a cell boundary falls on different sides in fp32
      and fp64;
  (c) the kernel reproduces, within the tolerances, the fp32 BUILD OF THE ORACLE run beside the fp64 one on the same inputs (a
      decision both fp32 evaluations take the same way and fp64 the other, e.g. the termination threshold on the base height):
      no bound applies — the kernel IS a valid fp32 evaluation of the restatement there;
  (b) precision: the CONDITIONING of the environment-step — the larger of (i) the error of the fp32 build of the oracle
      (oracle/_build/libgo1oracle32.so, the same restatement with real = float) run beside the fp64 one and (ii), in the full-step runs,
      what perturbing the fp64 oracle's inputs by one fp32 ulp does to its own result (ShadowPert, three draws) — uses up a twentieth of a
      tolerance (the median of (i) is 0.3 %) AND comes within a factor 25 of the kernel's error (RULE_B_*): the state is ill-conditioned in
      fp32 (deep interpenetration with kilonewton impulses, a non-converged 20-contact solve), whoever computes it.  (In bulk
      the kernel's error equals the fp32 oracle's — finish() prints the ratio of the medians, 0.8-1.0 on the MI355X, and the
      99 % quantiles coincide; in an ill-conditioned step the two are different draws from a heavy-tailed amplification of two
      different rounding sequences, hence the factor.)
tests/test_oracle_precision.py measures the rate at which the fp32 oracle alone leaves the tolerances: the same order.

The kernels the PRODUCT launches carry no signature code (csrc/go1sim.hip: go1_step_kernel / _hf / _walls; the `_sig` twins are
separate template instances).  test_product_instances_match_oracle runs those three at BASELINE's 4096 environments against the
oracle with rules (b) / (c); rule (a) is admitted there only for an environment-step in which the product instance's outputs
are BIT-IDENTICAL to its `_sig` twin's, stepped beside it from the same inputs — then, and only then, the twin's record
describes what the product instance computed.
"""
import os

import numpy as np
import pytest
import torch

import go1sim_host as H
from util import make_sim, randomize_dr, standing_state

pytestmark = pytest.mark.gpu

STATE_KEYS = ["root_states", "dof_pos", "dof_vel"]


def gpu_pair(variant, N, seed=3, **kw):
    import pyoracle
    cfg, S, meta, Bc = make_sim(variant, N, seed=seed, **kw)
    randomize_dr(Bc, seed)
    Bc.enable_contact_signature()
    orc = pyoracle.Oracle(S, Bc)
    orc.reset_idx()
    return cfg, S, meta, Bc, orc


class Shadow32:
    """the fp32 build of the oracle stepping beside the fp64 one from the same (re-synchronised) inputs"""

    def __init__(self, S, Bc, orc):
        import pyoracle
        self.Bc, self.orc = Bc, orc
        self.B = Bc.clone_to("cpu")
        self.o = pyoracle.Oracle(S, self.B, fp32=True)
        self.sync()

    def sync(self):
        for k, t in self.Bc.tensors.items():
            if t is not None and self.B.tensors.get(k) is not None:
                self.B.tensors[k].copy_(t)
        c, o = self.orc.ctr, self.o.ctr
        o.common_step_counter, o.lag_head, o.history_slot = c.common_step_counter, c.lag_head, c.history_slot


class ShadowPert:
    """Conditioning probe: K copies of the fp64 oracle stepping from the same (re-synchronised) inputs PERTURBED BY ONE FP32 ULP
    (root state, joint positions and rates times 1 + 2^-23 U(-1, 1), fixed seeds).  In a well-conditioned environment-step the outputs move
    by 1e-5 (1e-3 of a tolerance); where the step hides a discrete decision — a leg-leg contact between nearly parallel capsules, an
    impact on a speculative contact — they move by whole tolerances, whatever arithmetic evaluates it: measured on the two states the
    hardware kernel left a tolerance in with identical contact and active sets while the fp32 oracle stayed inside (hf run, env 2413
    step 12: 0.08 rad/s = x4 the joint-rate tolerance; env 2320 step 25: 0.011 rad/s base rate = x3.5), 2e-5 on an ordinary state."""

    def __init__(self, S, Bc, orc, draws=3, seed=77):
        import pyoracle
        self.Bc, self.orc, self.draws = Bc, orc, draws
        self.snap = Bc.clone_to("cpu")              # the pre-step state (ONE copy per step; the perturbed copies are made on demand)
        self.B = [Bc.clone_to("cpu") for _ in range(draws)]
        self.o = [pyoracle.Oracle(S, B) for B in self.B]
        self.g = torch.Generator().manual_seed(seed)
        self.pending, self.eval_cfg = None, None
        self.sync()

    def set_eval_config(self, S_eval, num_train):
        for o in self.o:
            o.set_eval_config(S_eval, num_train)

    def sync(self):
        for k, t in self.Bc.tensors.items():
            if t is not None and self.snap.tensors.get(k) is not None:
                self.snap.tensors[k].copy_(t)
        c = self.orc.ctr
        self.ctr = (c.common_step_counter, c.lag_head, c.history_slot)

    def step(self, a):
        """note the actions of this step; the perturbed oracles only run when Attribution.step() meets an environment the other rules
        leave unexplained"""
        self.pending = a

    def run(self):
        if self.pending is None:
            return
        for B, o in zip(self.B, self.o):
            for k, t in self.snap.tensors.items():
                if t is not None and B.tensors.get(k) is not None:
                    B.tensors[k].copy_(t)
            for k in ("root_states", "dof_pos", "dof_vel"):
                t = B.tensors[k]
                t.mul_(1.0 + (torch.rand(t.shape, generator=self.g) * 2.0 - 1.0) * 2.0 ** -23)
            o.ctr.common_step_counter, o.ctr.lag_head, o.ctr.history_slot = self.ctr
            o.step(self.pending)
        self.pending = None


def to_gpu(S, Bc, product=False):
    """product: WITHOUT the contact-signature buffer, so that go1sim_step launches the instances the product launches
    (go1_step_kernel / _hf / _walls, csrc/go1sim.hip `launch`) instead of their `_sig` twins"""
    Bg = Bc.clone_to("cuda:0")
    if product:
        Bg.tensors["contact_signature"] = None
        Bg.refresh_struct()
        assert not Bg.struct.contact_signature
    sim = H.Go1Sim(S, Bg, 0)
    return Bg, sim


ROW_MAJOR = ("obs_buf", "privileged_obs_buf", "obs_history")


def identical_envs(Ba, Bb, N, keys, tally=None):
    """(N,) bool: every listed output of the environment is bit-identical in the two buffer sets
    (tally: dict key -> environment-steps in which THAT output differs, accumulated)"""
    same = torch.ones(N, dtype=torch.bool)
    for k in keys:
        a, b = Ba.tensors[k], Bb.tensors[k]
        ne = a != b
        if a.is_floating_point():
            ne = ne & ~(a.isnan() & b.isnan())
        per_env = ne.reshape(N, -1).any(1) if k in ROW_MAJOR else ne.reshape(-1, N).any(0)
        same &= ~per_env.cpu()
        if tally is not None and bool(per_env.any()):
            tally[k] = tally.get(k, 0) + int(per_env.sum())
    return same


class ProductPair:
    """a product instance of the step kernel (no signature buffer) and its `_sig` twin, stepped from the same inputs"""

    def __init__(self, S, Bc, orc, keys):
        self.Bc, self.orc, self.keys = Bc, orc, list(keys) + ["contact_forces", "reset_buf", "time_out_buf"]
        self.Bg, self.sim = to_gpu(S, Bc, product=True)
        self.Bt, self.sim_t = to_gpu(S, Bc)
        self.env_steps = self.differ = 0
        self.by_key = {}

    def step(self, a):
        self.sim.step(a)
        self.sim_t.step(a)
        torch.cuda.synchronize()
        same = identical_envs(self.Bg, self.Bt, self.Bc.root_states.shape[1], self.keys, self.by_key)
        self.env_steps += same.numel()
        self.differ += int((~same).sum())
        return (self.Bt, same)

    def sync(self):
        sync_from(self.Bc, self.Bg, self.sim, self.orc)
        sync_from(self.Bc, self.Bt, self.sim_t, self.orc)

    def note(self):
        by = (" (by output: " + ", ".join(f"{k} {v}" for k, v in self.by_key.items()) + ")") if self.by_key else ""
        return f"; product instance vs `_sig` twin: {self.differ} of {self.env_steps} env-steps not bit-identical{by}"


def sync_from(Bc, Bg, sim, orc):
    for k, t in Bc.tensors.items():
        if t is not None and k in Bg.tensors and Bg.tensors[k] is not None:
            Bg.tensors[k].copy_(t)
    sim.set_counters(orc.ctr.common_step_counter, orc.ctr.lag_head)


ATTRIBUTED_BOUND = 50.0          # x tolerance: how far a step attributed to fp32 precision alone (rule (b)) may be off.  Measured worst: x10.8 in
                                 # round 3's 256-environment runs, x32.7 (fp32 oracle x1.3 itself) in 163,840 env-steps on the relief at 4096 environments
# rule (b): the fp32 oracle's own error is >= RULE_B_FLOOR of a tolerance (its median is 0.003, its 99 % quantile 0.015-0.1: such a step is
# 15-30 x less well conditioned than the bulk) AND the kernel's error is within RULE_B_FACTOR of it.  Measured on the MI355X (round 4,
# 4096 envs x 40 steps per instance, profiles/r04_parity_rates.txt): flat terrain 0 of 163,840 env-steps outside the tolerances at all;
# relief: the widest spread between the two fp32 evaluations in a step the kernel left the tolerances in was x20.8 (kernel 1.89 x the
# dof_vel tolerance, fp32 oracle 0.091 x, identical contact and active sets) — round 3's 256-env runs had seen x10.8.
RULE_B_FLOOR, RULE_B_FACTOR = 0.05, 25.0
# FROZEN in round 5 (the review of round 4): ATTRIBUTED_*, RULE_B_* keep the values measured on the list-order solver; the switch of the sweep
# order was run against them unchanged.  Added in round 5, per rule (advisor finding: rule (a) had no bound and no budget of its own):
RULE_A_BOUND = 500.0             # x tolerance for a step attributed to a different contact list / active set that the fp32 oracle does NOT share
                                 # with the kernel (rule (c) has no bound by construction).  Measured worst: x221 (4096 envs on the relief,
                                 # a contact impulse of kilonewtons x 5 ms entering / leaving the list; the fp32 oracle's own worst there x425)
RULE_A_ACTIVE_RATE = 4e-3        # env-steps whose LISTS agree and whose active set after some sweep differs: where a solver regression would hide
RULE_BC_RATE = 4e-3              # env-steps attributed to precision (b: fp32 oracle / perturbation probe, c: kernel == fp32 oracle).  Measured worst:
                                 # 6 of 2048 = 2.9e-3, all rule (c), robots tumbling into risers (gpurun call r5a; a first value of 2.5e-3 failed on it)
ATTRIBUTED_RATE = 5e-3           # fraction of environment-steps allowed to be attributed (flat terrain, measured: 0; robots thrown INTO a
                                 # staircase with kilonewton depenetration impulses: 2.4e-3; fp32-vs-fp64 oracle alone: 3e-5 .. 2e-3)


class Attribution:
    """Per-step bookkeeping of the environments outside the tolerances: every one must differ in its contact signature."""

    def __init__(self, N, residual=(0, 0.0)):
        """residual = (count, bound): at most `count` environment-steps of the whole run may stay UNEXPLAINED — outside a tolerance by at
        most `bound` x, none of the rules applying — instead of failing at the first one.  (0, 0) in every test of this file: nothing
        unexplained is admitted (the mechanism exists for investigations: tools/debug/hf_env_replay.py grew out of one)."""
        self.N, self.env_steps, self.bad, self.attributed, self.worst_ratio, self.worst_unattr, self.worst_a = N, 0, 0, 0, 0.0, 0.0, 0.0
        self.r_all, self.r32_all = [], []
        self.note = ""
        self.residual, self.unexplained = residual, 0
        self.by_rule = {k: 0 for k in ("c", "a-list", "a-active", "local", "b-fp32", "b-pert")}
        self.worst_by_rule = {k: 0.0 for k in self.by_rule}

    def _sig_words(self, Bc, Bx, words):
        """(N,) bool: the contact signatures differ in one of the listed words (include/go1sim.h GO1_SIG_WORDS) of some substep"""
        a, b = Bx.contact_signature.cpu(), Bc.contact_signature
        rows = [r for r in range(a.shape[0]) if r % 4 in words]
        return (a[rows] != b[rows]).any(0)

    def ratio(self, a, b, atol, rtol=0.0, env_dim=-1):
        dev = "cuda" if (a.is_cuda or b.is_cuda) else "cpu"          # (the arithmetic of the CHECK runs where the data is: 4096-env histories)
        a, b = a.to(dev).double(), b.to(dev).double()
        r = (a - b).abs() / (atol + rtol * b.abs())
        r = torch.nan_to_num(r, nan=float("inf"))
        if env_dim == 0:
            return r.reshape(self.N, -1).max(1).values.cpu()
        return r.reshape(-1, self.N).max(0).values.cpu()

    def step(self, ratio_fn, Bg, Bc, B32=None, reset_key=None, also_attributed=None, twin=None, pert=None):
        """ratio_fn(Bx, Bref) -> (N,) worst error / tolerance of every environment this step; B32: the fp32 oracle's buffers;
        reset_key: a buffer whose mismatch (termination decided differently) puts the environment outside the tolerances;
        twin = (Bt, identical): Bg carries no signature (a product instance) — rule (a) reads the record of the `_sig` twin Bt for
        the environments whose outputs are bit-identical in the two (`identical`, (N,) bool)."""
        ratio = ratio_fn(Bg, Bc)
        if twin is not None:
            assert Bg.contact_signature is None
            sig = (twin[0].contact_signature.cpu() != Bc.contact_signature).any(0) & twin[1]
        else:
            sig = (Bg.contact_signature.cpu() != Bc.contact_signature).any(0)
        bad = ratio > 1.0
        if reset_key is not None:
            bad = bad | (Bg.tensors[reset_key].cpu().bool() != Bc.tensors[reset_key].bool())
        same32 = torch.zeros_like(bad)
        sig_raw = sig.clone()                      # the contact signature itself
        sig_list = self._sig_words(Bc, twin[0] if twin is not None else Bg, (0, 1, 2)) & sig_raw       # ... in the LISTED points / pairs / limit legs
        if also_attributed is not None:            # a test-specific, stated rule (e.g. height-scan samples on a cell boundary)
            sig = sig | also_attributed
        sig_a = sig.clone()                        # rule (a): kernel and oracle solved DIFFERENT discrete problems this step
        rule_b32 = rule_bp = torch.zeros_like(bad)
        if B32 is not None:
            # (b): ill-conditioned in fp32 — the fp32 oracle, whose error in a well-conditioned environment-step is 1-3 % of a
            # tolerance (printed by finish()), uses up RULE_B_FLOOR of it here AND is within RULE_B_FACTOR of the kernel's error;
            # (c): the kernel REPRODUCES the fp32 oracle within the tolerances (a decision both fp32 evaluations take the same
            # way and fp64 the other — e.g. a termination threshold): no bound on how far that is from the fp64 result
            ratio32 = ratio_fn(B32, Bc)
            same32_pre = ratio_fn(Bg, B32) <= 1.0
            rule_b32 = (ratio32 > RULE_B_FLOOR) & (ratio <= RULE_B_FACTOR * ratio32)
            need = bad & ~sig & ~(rule_b32 | same32_pre)
            if pert is not None and bool(need.any()):     # conditioning of the step itself (ShadowPert): the larger of the fp32 oracle's error and
                pert.run()                                 # of what one-ulp input perturbations do to the fp64 oracle's own result
                for Bp in pert.B:
                    rp = ratio_fn(Bp, Bc)
                    if reset_key is not None:      # (a termination decided differently under the perturbation: a threshold sits here)
                        rp = torch.where(Bp.tensors[reset_key].bool() != Bc.tensors[reset_key].bool(), torch.full_like(rp, 1e3), rp)
                    ratio32 = torch.maximum(ratio32, rp)
            same32 = ratio_fn(Bg, B32) <= 1.0
            if reset_key is not None:
                same32 = same32 & (Bg.tensors[reset_key].cpu().bool() == B32.tensors[reset_key].bool())
            rule_bp = (ratio32 > RULE_B_FLOOR) & (ratio <= RULE_B_FACTOR * ratio32) & ~rule_b32
            sig = sig | rule_b32 | rule_bp | same32
            self.r32_all.append(ratio32.clone()); self.r_all.append(ratio.clone())
        # who carries what (every out-of-tolerance environment-step is counted under the FIRST rule that explains it, in this order):
        #   c: the kernel reproduces the fp32 oracle within the tolerances (the strongest statement: the kernel IS a valid fp32 evaluation there,
        #   whatever fp64 decides — gpurun call r5c: a x2923 step on the relief where kernel and fp32 oracle agree to 1e-5 of it);
        #   a-list: the listed contact points / self pairs / limit-row legs differ; a-active: the lists agree, the active set after some sweep
        #   (or a restitution branch) differs; local: the test's own stated rule; b-fp32: the fp32
        #   oracle's own error explains it; b-pert: only the one-ulp perturbation probe of the fp64 oracle does
        left = bad.clone()
        for name, mask in (("c", same32), ("a-list", sig_list), ("a-active", sig_raw), ("local", sig_a), ("b-fp32", rule_b32), ("b-pert", rule_bp)):
            hit = left & mask
            self.by_rule[name] += int(hit.sum())
            if bool(hit.any()):
                self.worst_by_rule[name] = max(self.worst_by_rule[name], float(ratio[hit].max()))
            left = left & ~mask
        un = bad & ~sig
        if bool(un.any()) and B32 is not None:
            print("UNATTRIBUTED", [(int(e), round(float(ratio[e]), 2), round(float(ratio32[e]), 3)) for e in un.nonzero().flatten()[:8]],
                  getattr(ratio_fn, "detail", lambda *_: "")(Bg, Bc, un))
            for e in un.nonzero().flatten()[:2].tolist():          # the state the disagreement happened in (for the record in the log)
                f = lambda t: [round(float(x), 4) for x in t]
                print(f"  env {e}: oracle root {f(Bc.root_states[:, e])}\n    q {f(Bc.dof_pos[:, e])}\n    qd oracle {f(Bc.dof_vel[:, e])}\n"
                      f"    qd kernel - oracle {f(Bg.dof_vel[:, e].cpu() - Bc.dof_vel[:, e])}\n    qd fp32 oracle - oracle {f(B32.dof_vel[:, e] - Bc.dof_vel[:, e])}\n"
                      f"    contact forces (oracle, z of 17 bodies) {f(Bc.contact_forces.view(17, 3, -1)[:, 2, e])}\n"
                      f"    signature {[hex(int(x) & 0xffffffff) for x in Bc.contact_signature[:, e]]}")
                for k in ("episode_sums", "command_sums", "episode_sums_eval", "rew_buf", "commands"):
                    tg, tc, t3 = Bg.tensors.get(k), Bc.tensors.get(k), B32.tensors.get(k)
                    if tg is None or tc is None or tg.shape[-1] != self.N:
                        continue
                    dg = (tg.reshape(-1, self.N)[:, e].cpu().double() - tc.reshape(-1, self.N)[:, e].double())
                    rows = (dg.abs() > 1e-4).nonzero().flatten().tolist()[:6]
                    if rows:
                        print(f"    {k}: rows {rows} kernel {f(tg.reshape(-1, self.N)[rows, e])} oracle {f(tc.reshape(-1, self.N)[rows, e])} fp32 {f(t3.reshape(-1, self.N)[rows, e])}"
                              f" reset k/o {int(Bg.reset_buf[e])}/{int(Bc.reset_buf[e])} len {int(Bc.episode_length_buf[e])}")
        self.env_steps += self.N
        self.bad += int(bad.sum()); self.attributed += int((bad & sig).sum())
        # ATTRIBUTED_BOUND applies to what is attributed to PRECISION alone (rule (b)): a contact point entering / leaving the list or the
        # active set (rule (a)) changes the step by a whole contact impulse — kilonewtons x 5 ms for a robot thrown into a riser; measured
        # at 4096 environments on the relief: up to x221 of a tolerance, the fp32 oracle's own worst x425 — and rule (c) has no bound by
        # construction.  The RATE of all of them stays bounded (ATTRIBUTED_RATE).
        bounded = bad & sig & ~sig_a & ~same32
        if bool(bounded.any()):
            self.worst_ratio = max(self.worst_ratio, float(ratio[bounded].max()))
        in_a = bad & sig_a & ~same32
        if bool(in_a.any()):
            self.worst_a = max(self.worst_a, float(ratio[in_a].max()))
        if bool(un.any()):
            self.worst_unattr = max(self.worst_unattr, float(ratio[un].max()))
        if bool(un.any()):
            self.unexplained += int(un.sum())
            assert self.unexplained <= self.residual[0] and float(ratio[un].max()) <= self.residual[1], \
                f"environments {un.nonzero().flatten().tolist()[:8]} exceed the tolerances (worst x{float(ratio[un].max()):.1f}) with identical contact sets"
        return bad

    def finish(self, what):
        rate = self.attributed / max(self.env_steps, 1)
        q = ""
        if self.r_all:
            r, r32 = torch.cat(self.r_all), torch.cat(self.r32_all)
            q = (f"; error / tolerance, median | 99 % | max: kernel {float(r.median()):.3f} | {float(r.quantile(0.99)):.3f} | {float(r.max()):.2f}, "
                 f"fp32 oracle {float(r32.median()):.3f} | {float(r32.quantile(0.99)):.3f} | {float(r32.max()):.2f}; "
                 f"bulk factor kernel / fp32 oracle (ratio of medians) {float(r.median()) / max(float(r32.median()), 1e-9):.1f}")
        res = "all attributed" if self.unexplained == 0 else \
            f"{self.attributed} attributed, {self.unexplained} UNEXPLAINED (worst x{self.worst_unattr:.2f} of the tolerance; allowed: {self.residual[0]} up to x{self.residual[1]})"
        split = ", ".join(f"{k} {v} (x{self.worst_by_rule[k]:.1f})" for k, v in self.by_rule.items() if v)
        line = (f"{what}: {self.env_steps} env-steps, {self.bad} outside the tolerances, {res} "
                f"(rate {rate:.2e}, worst x{self.worst_ratio:.1f} of the tolerance by precision alone{'; by rule: ' + split if split else ''}){q}{self.note}")
        print(line)
        if os.environ.get("GO1_PARITY_LOG"):            # the GPU run's summaries, committed as profiles/r04_parity_rates.txt
            with open(os.environ["GO1_PARITY_LOG"], "a") as f:
                f.write(line + "\n")
        assert rate <= ATTRIBUTED_RATE, rate
        assert self.worst_ratio <= ATTRIBUTED_BOUND, self.worst_ratio
        # rule (a) has its own bound and the active-set-only part of it its own rate (a solver regression would show up THERE: same lists,
        # another active set); rules (b) + (c) — precision — their own rate
        assert self.worst_a <= RULE_A_BOUND, self.worst_a
        n = max(self.env_steps, 1)
        assert self.by_rule["a-active"] / n <= RULE_A_ACTIVE_RATE, (self.by_rule, n)
        assert (self.by_rule["b-fp32"] + self.by_rule["b-pert"] + self.by_rule["c"]) / n <= RULE_BC_RATE, (self.by_rule, n)


def grazing_collision_count(att, Bg, Bc, keys, rows=()):
    """test-local rule shared by the full-step tests (counted as `local` in the per-rule split).  _reward_collision counts the penalised
    bodies whose contact force exceeds 0.1 N (corl_rewards.py:70-72): a body grazing the ground with a force within 0.1 N of that threshold
    is counted by one evaluation and not by the other — a discrete flip the contact signature does not record (the contact is listed and
    pressing in both); it shows in the raw per-term running sums ONLY: everything else of the environment must be inside its tolerance."""
    N = att.N
    def grazing(B):
        f = B.contact_forces.view(17, 3, N).cpu().norm(dim=1)
        return ((f > 0) & (f < 0.2)).any(0)
    sums = ("episode_sums", "episode_sums_eval", "command_sums")
    rest_ok = make_ratio(att, tuple(k for k in keys if k[0] not in sums), rows)(Bg, Bc) <= 1.0
    return (grazing(Bg) | grazing(Bc)) & rest_ok


SUBSTEP_TOL = (("root_states", 2e-4, 1e-4), ("dof_pos", 2e-4, 1e-4), ("dof_vel", 3e-3, 1e-4), ("contact_forces", 5e-2, 2e-3))


def make_ratio(att, keys, rows=()):
    """ratio_fn for Attribution.step over [C][N] quantities `keys` and (N, K) row-major ones `rows`: (name, atol, rtol)"""
    def fn(Bx, Bref):
        ratio = torch.zeros(att.N, dtype=torch.float64)
        for k, tol, rt in keys:
            ratio = torch.maximum(ratio, att.ratio(Bx.tensors[k], Bref.tensors[k], tol, rt))
        for k, tol, rt in rows:
            ratio = torch.maximum(ratio, att.ratio(Bx.tensors[k], Bref.tensors[k], tol, rt, env_dim=0))
        return ratio

    def detail(Bx, Bref, mask):
        out = {}
        for k, tol, rt in keys:
            out[k] = round(float(att.ratio(Bx.tensors[k], Bref.tensors[k], tol, rt)[mask].max()), 2)
        for k, tol, rt in rows:
            out[k] = round(float(att.ratio(Bx.tensors[k], Bref.tensors[k], tol, rt, env_dim=0)[mask].max()), 2)
        return {k: v for k, v in out.items() if v > 0.5}
    fn.detail = detail
    return fn


def frac_bad(a, b, atol, rtol=0.0):
    a, b = a.double().cpu(), b.double().cpu()
    bad = (a - b).abs() > atol + rtol * b.abs()
    dims = tuple(range(a.dim()))
    return bad, float(bad.float().mean())


@pytest.mark.parametrize("variant", ["train", "alt"])
def test_torque_model_matches_oracle(variant):
    N = 256
    cfg, S, meta, Bc, orc = gpu_pair(variant, N)
    g = torch.Generator().manual_seed(0)
    Bg, sim = to_gpu(S, Bc)
    for step in range(9):
        q = torch.tensor(list(S.default_dof_pos)).unsqueeze(1) + torch.empty(12, N).uniform_(-0.8, 0.8, generator=g)
        qd = torch.empty(12, N).uniform_(-10, 10, generator=g)
        a = torch.empty(12, N).uniform_(-4, 4, generator=g)
        for B in (Bc, Bg):
            B.dof_pos.copy_(q); B.dof_vel.copy_(qd)
        orc.compute_torques(a.numpy())
        sim.compute_torques(a.cuda().contiguous())
        torch.cuda.synchronize()
        np.testing.assert_allclose(Bg.torques.cpu().numpy(), Bc.torques.numpy(), rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(Bg.joint_pos_target.cpu().numpy(), Bc.joint_pos_target.numpy(), rtol=1e-6, atol=1e-6)
    for k in ("joint_pos_err_last", "joint_pos_err_last_last", "joint_vel_last", "joint_vel_last_last", "lag_buffer"):
        np.testing.assert_allclose(Bg.tensors[k].cpu().numpy(), Bc.tensors[k].numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("scenario", ["flight", "standing", "dropped", "tumbling"])
def test_physics_substep_matches_oracle(scenario):
    N = 256
    cfg, S, meta, Bc, orc = gpu_pair("train", N, extra={"domain_rand": dict(randomize_gravity=False)})
    g = torch.Generator().manual_seed(1)
    if scenario == "flight":
        Bc.root_states[2] = 2.0
        Bc.dof_vel.uniform_(-5, 5, generator=g)
        Bc.root_states[7:13].uniform_(-2, 2, generator=g)
    elif scenario == "standing":
        standing_state(S, Bc, z=0.28)
    elif scenario == "dropped":
        Bc.root_states[2].uniform_(0.05, 0.3, generator=g)      # some start interpenetrating: depenetration path
        Bc.root_states[9] = -1.5                                # restitution path
    elif scenario == "tumbling":
        q = torch.randn(4, N, generator=g)
        Bc.root_states[3:7] = q / q.norm(dim=0, keepdim=True)
        Bc.root_states[2].uniform_(0.08, 0.35, generator=g)
        Bc.root_states[7:13].uniform_(-2, 2, generator=g)
        Bc.dof_vel.uniform_(-5, 5, generator=g)
    Bc.torques.uniform_(-20, 20, generator=g)
    Bg, sim = to_gpu(S, Bc)
    sh = Shadow32(S, Bc, orc)
    att = Attribution(N)
    for it in range(6):
        orc.physics_substep()
        sh.o.physics_substep()
        sim.physics_substep()
        torch.cuda.synchronize()
        assert torch.isfinite(Bg.root_states).all() and torch.isfinite(Bg.dof_vel).all()
        att.step(make_ratio(att, SUBSTEP_TOL), Bg, Bc, sh.B)
        sync_from(Bc, Bg, sim, orc)      # re-synchronise so that one substep is compared at a time
        sh.sync()
    att.finish(f"substep[{scenario}]")
    if scenario in ("flight", "standing"):
        assert att.bad == 0
    if scenario != "flight":
        assert float(Bc.contact_forces.abs().max()) > 1.0


FULL_STEP_TOL = (("root_states", 1e-3, 1e-3), ("dof_pos", 1e-3, 0), ("dof_vel", 2e-2, 1e-3), ("rew_buf", 2e-4, 1e-3),
                 ("commands", 1e-5, 0), ("gait_indices", 1e-5, 0), ("clock_inputs", 1e-4, 0),
                 ("desired_contact_states", 1e-4, 0), ("torques", 5e-3, 1e-3), ("foot_positions", 1e-3, 0),
                 ("episode_sums", 1e-3, 1e-3), ("command_sums", 1e-3, 1e-3), ("motor_offsets", 1e-6, 0),
                 ("motor_strengths", 1e-6, 0), ("last_actions", 1e-6, 0),
                 # state the torque model carries (LDS stash in the step kernel, written back once per step)
                 ("joint_pos_err_last", 1e-3, 0), ("joint_pos_err_last_last", 1e-3, 0), ("joint_vel_last", 2e-2, 1e-3),
                 ("joint_vel_last_last", 2e-2, 1e-3), ("joint_pos_target", 1e-5, 0), ("lag_buffer", 1e-5, 0))
ROW_TOL = (("obs_buf", 3e-3, 1e-3), ("privileged_obs_buf", 1e-5, 0), ("obs_history", 3e-3, 1e-3))


# Cross-check of the frozen attribution constants (the review of round 4, item 7): the `-alt` cases of test_product_instances_match_oracle (suite
# members since round 6; GO1_PARITY_ALT=1 still switches every product run over) re-run the 4096-environment product-instance
# tests with ANOTHER seed for states / domain randomisation / action stream and ANOTHER relief (rough_field seed) — same constants, same rules.
# The summaries of both settings are committed side by side (profiles/r05_parity_rates.txt, r05_parity_rates_alt_seed.txt).
PARITY_ALT = os.environ.get("GO1_PARITY_ALT", "0") == "1"


def run_full_step_comparison(variant, N, steps, seed=11, prepare=None, watch=None, what="full step", product=False, alt=None):
    """HIP step vs oracle step on identical state / action / RNG streams, re-synchronised after every step so that each
    step is compared on its own (a free-running pair diverges through contact-mode flips, as two fp32 PhysX runs
    would).  Every environment outside the per-quantity tolerances must be attributed (module docstring); returns the
    Attribution record and event counts."""
    alt = (PARITY_ALT if alt is None else alt) and product
    if alt:
        seed, what = seed + 1000, what + " (alt seed)"
    cfg, S, meta, Bc, orc = gpu_pair(variant, N, seed=seed)
    pp = ProductPair(S, Bc, orc, [k for k, _, _ in FULL_STEP_TOL + ROW_TOL]) if product else None
    Bg, sim = (pp.Bg, pp.sim) if product else to_gpu(S, Bc)
    rng = np.random.default_rng(1000 if alt else 0)
    Bc.episode_length_buf[:] = torch.randint(0, S.max_episode_length, (N,), dtype=torch.int32, generator=torch.Generator().manual_seed(2))
    if prepare is not None:
        prepare(S, Bc)
    pp.sync() if product else sync_from(Bc, Bg, sim, orc)
    sh = Shadow32(S, Bc, orc)
    sp = ShadowPert(S, Bc, orc) if product else None         # (the conditioning probe: the 4096-environment runs and the relief)
    resets = 0
    resamples = 0
    timeouts = 0
    att = Attribution(N)
    for step in range(steps):
        a = (rng.standard_normal((N, 12)) * (1.0 if step % 2 else 0.3)).astype(np.float32)
        if step == 20:
            a[:] = 12.0          # action clipping
        cmd_before = Bc.commands.clone()
        if watch is not None:
            watch(S, Bc, "before")
        orc.step(a)
        sh.o.step(a)
        if sp is not None:
            sp.step(a)
        twin = None
        if product:
            twin = pp.step(torch.from_numpy(a).cuda())
        else:
            sim.step(torch.from_numpy(a).cuda())
            torch.cuda.synchronize()
        cpu_reset = Bc.reset_buf.bool()
        np.testing.assert_array_equal(Bg.time_out_buf.cpu().numpy(), Bc.time_out_buf.numpy())
        bad_env = att.step(make_ratio(att, FULL_STEP_TOL, ROW_TOL), Bg, Bc, sh.B, reset_key="reset_buf", twin=twin, pert=sp,
                           also_attributed=grazing_collision_count(att, Bg, Bc, FULL_STEP_TOL, ROW_TOL))
        timeouts += int(Bc.time_out_buf.sum())
        np.testing.assert_array_equal(Bg.env_command_bins.cpu().numpy()[~bad_env.numpy()], Bc.env_command_bins.numpy()[~bad_env.numpy()])
        np.testing.assert_allclose(Bg.curriculum_weights.cpu().numpy(), Bc.curriculum_weights.numpy(), atol=0.21 * float(bad_env.sum()) + 1e-6)
        resets += int(cpu_reset.sum())
        resamples += int((cmd_before != Bc.commands).any(0).sum())
        if watch is not None:
            watch(S, Bc, "after")
            for k in ("payloads", "friction_coeffs", "restitutions", "com_displacements"):
                bad, _ = frac_bad(Bg.tensors[k], Bc.tensors[k], 1e-6)
                assert not bool(bad[..., ~bad_env].any()), k
        pp.sync() if product else sync_from(Bc, Bg, sim, orc)
        sh.sync()
        if sp is not None:
            sp.sync()
    assert int(Bg.fault_counts[:10].sum()) == 0, Bg.fault_counts.tolist()
    att.note = pp.note() if product else ""
    assert not product or pp.differ == 0, pp.note()        # (the twin relation: see test_product_instances_match_oracle)
    att.finish(f"{what} [{variant}, {N} envs x {steps} steps]")
    return att, resets, resamples, timeouts


@pytest.mark.parametrize("variant", ["train_noise", "alt"])
def test_full_step_matches_oracle(variant):
    att, resets, resamples, _ = run_full_step_comparison(variant, 512, 40)
    assert resets > 20 and resamples > 20      # resets and interval resamples were exercised


@pytest.mark.parametrize("variant,N", [("train_noise", 1), ("dr", 21)])
def test_ragged_env_counts_match_oracle(variant, N):
    """environment counts that do not fill a workgroup of 16: scripts/play.py's single environment (play.py:62) and a ragged
    second workgroup."""
    run_full_step_comparison(variant, N, 12, what="ragged")


@pytest.mark.parametrize("case", range(6))
def test_full_step_under_random_configurations(case):
    """configuration fuzz on the hardware (fixed seeds; the generator of tests/golden/variants.py, the same that drives the emulated
    kernel's fuzz in tests/test_emu_parity.py): random consistent sets of observation / privileged-observation / controller /
    reward / termination / command switches, HIP step vs oracle on identical streams, re-synchronised every step."""
    from golden.variants import random_switches
    rng = np.random.default_rng(1000 + case)
    extra = random_switches(rng)
    N, steps = 128, 6
    cfg, S, meta, Bc, orc = gpu_pair("train_noise", N, seed=40 + case, extra=extra)
    Bg, sim = to_gpu(S, Bc)
    Bc.episode_length_buf[:] = torch.randint(0, int(S.max_episode_length), (N,), dtype=torch.int32, generator=torch.Generator().manual_seed(case))
    sync_from(Bc, Bg, sim, orc)
    sh = Shadow32(S, Bc, orc)
    att = Attribution(N)
    for step in range(steps):
        a = (rng.standard_normal((N, 12)) * (2.0 if step == 1 else 0.5)).astype(np.float32)
        orc.step(a)
        sh.o.step(a)
        sim.step(torch.from_numpy(a).cuda())
        torch.cuda.synchronize()
        keys = (("root_states", 1e-3, 1e-3), ("dof_pos", 1e-3, 0), ("dof_vel", 2e-2, 1e-3), ("rew_buf", 2e-4, 1e-3),
                ("commands", 1e-5, 0), ("torques", 5e-3, 1e-3), ("episode_sums", 1e-3, 1e-3), ("command_sums", 1e-3, 1e-3))
        rows = (("obs_buf", 3e-3, 1e-3), ("privileged_obs_buf", 3e-3, 1e-3))
        att.step(make_ratio(att, keys, rows), Bg, Bc, sh.B, reset_key="reset_buf")
        sync_from(Bc, Bg, sim, orc)
        sh.sync()
    att.finish(f"fuzz case {case}")
    assert int(Bg.fault_counts[:10].sum()) == 0


def test_push_teleport_and_rigid_rerandomisation_match_oracle():
    """The step-callback branches train.py leaves switched off — velocity pushes (north_star's "domain-randomisation
    pushes", legged_robot.py:1017-1026), edge teleport (:1028-1051) and re-drawn rigid-body properties
    (:706-708, 166-168) — all enabled, kernel vs oracle on identical streams, and each observed to fire."""
    seen = {"teleport": 0, "push": 0, "rigid": 0}
    keep = {}

    def prepare(S, B):
        N = B.root_states.shape[1]
        span_x, span_y = S.terrain_length * S.terrain_num_rows, S.terrain_width * S.terrain_num_cols
        g = torch.Generator().manual_seed(5)
        B.env_origins[0] = torch.where(torch.rand(N, generator=g) < 0.5, torch.full((N,), 0.45), torch.full((N,), span_x - 0.45))
        B.env_origins[1] = torch.where(torch.rand(N, generator=g) < 0.5, torch.full((N,), 0.45), torch.full((N,), span_y - 0.45))
        B.root_states[0] = B.env_origins[0] + torch.empty(N).uniform_(-0.2, 0.2, generator=g)
        B.root_states[1] = B.env_origins[1] + torch.empty(N).uniform_(-0.2, 0.2, generator=g)

    def watch(S, B, when):
        if when == "before":
            keep["xy"], keep["v"], keep["m"] = B.root_states[:2].clone(), B.root_states[7:9].clone(), B.payloads.clone()
            keep["len"] = B.episode_length_buf.clone()
            return
        live = B.reset_buf == 0
        jump = (B.root_states[:2] - keep["xy"]).abs().max(0).values
        seen["teleport"] += int(((jump > 50.0) & live).sum())
        pushed = live & ((keep["len"] + 1) % S.push_interval == 0)
        seen["push"] += int(pushed.sum())
        if bool(pushed.any()):
            assert float(B.root_states[7:9][:, pushed].abs().max()) <= S.max_push_vel_xy + 1e-6
        seen["rigid"] += int(((B.payloads != keep["m"]) & live).sum())
    run_full_step_comparison("dr", 256, 60, seed=17, prepare=prepare, watch=watch, what="callbacks")
    assert seen["teleport"] > 10 and seen["push"] > 50 and seen["rigid"] > 50, seen


def test_thousand_steps_match_oracle():
    """1000 consecutive policy steps (4000 physics substeps, a full episode length: time-outs, gravity
    impulses, DR refreshes, curriculum updates and command resampling all occur) of the HIP kernel against the oracle
    on identical streams, per-step tolerances of test_full_step_matches_oracle."""
    att, resets, resamples, timeouts = run_full_step_comparison("train_noise", 128, 1000, seed=5, what="episode")
    assert resets > 100 and resamples > 100 and timeouts > 0, (resets, resamples, timeouts)




def test_free_running_distributions_match_oracle():
    """No re-synchronisation: kernel and oracle start from the same state and receive the same action and RNG streams for
    700 policy steps (512 envs, N(0,1) actions, train.py configuration).  Individual trajectories separate at the first
    contact-mode flip — as two fp32 PhysX runs would — so the comparison is distributional: mean episode length, mean
    step reward, per-term episode sums per episode, foot-contact duty factor, mean base height, command-curriculum mass.
    ~600 episodes per side: sampling noise is ~5 % (1 sigma) on the episode statistics; bounds are ~4 sigma."""
    N, steps = 512, 700
    cfg, S, meta, Bc, orc = gpu_pair("train_noise", N, seed=29)
    Bg, sim = to_gpu(S, Bc)
    sync_from(Bc, Bg, sim, orc)
    Bg.episode_log.zero_()               # (the copy carries the oracle's initial reset_idx: the kernel accumulates, the oracle restarts)
    rng = np.random.default_rng(3)
    nr = S.num_rewards
    acc = {"cpu": np.zeros(5), "gpu": np.zeros(5)}
    logs = {"cpu": np.zeros(nr + 2), "gpu": np.zeros(nr + 2)}
    for step in range(steps):
        a = rng.standard_normal((N, 12)).astype(np.float32)
        orc.step(a)
        sim.step(torch.from_numpy(a).cuda())
        for tag, B in (("cpu", Bc), ("gpu", Bg)):
            logs[tag] += B.episode_log.cpu().double().numpy()      # the oracle restarts this sum every step, the kernel accumulates
            cf = B.contact_forces.view(17, 3, N)
            acc[tag] += np.array([float(B.reset_buf.sum()), float(B.rew_buf.sum()), float((cf[[4, 8, 12, 16], 2] > 1.0).float().mean()),
                                  float(B.root_states[2].mean()), float(B.time_out_buf.sum())])
        Bg.episode_log.zero_()
    torch.cuda.synchronize()
    out = {}
    for tag, B in (("cpu", Bc), ("gpu", Bg)):
        log = logs[tag]
        n_ep = max(log[nr + 1], 1.0)
        out[tag] = dict(ep_len=N * steps / max(acc[tag][0], 1), rew=acc[tag][1] / (N * steps), duty=acc[tag][2] / steps,
                        height=acc[tag][3] / steps, terms=log[:nr + 1] / n_ep, weight_mass=float(B.curriculum_weights.sum()))
    c, g = out["cpu"], out["gpu"]
    print(f"free-running: episode length {c['ep_len']:.1f} / {g['ep_len']:.1f}, step reward {c['rew']:.5f} / {g['rew']:.5f}, "
          f"foot duty {c['duty']:.3f} / {g['duty']:.3f}, base height {c['height']:.4f} / {g['height']:.4f} (oracle / HIP)")
    assert acc["cpu"][0] > 300 and acc["gpu"][0] > 300
    assert abs(c["ep_len"] - g["ep_len"]) <= 0.2 * c["ep_len"]
    assert abs(c["duty"] - g["duty"]) <= 0.03 and abs(c["height"] - g["height"]) <= 0.01
    assert abs(c["rew"] - g["rew"]) <= 0.2 * abs(c["rew"]) + 2e-4
    scale = np.abs(c["terms"]).max()
    assert np.all(np.abs(c["terms"] - g["terms"]) <= 0.25 * np.abs(c["terms"]) + 0.03 * scale), (c["terms"], g["terms"])
    assert abs(c["weight_mass"] - g["weight_mass"]) <= 0.05 * c["weight_mass"] + 1.0
    assert int(Bg.fault_counts[:10].sum()) == 0


def rough_field(rows=240, cols=240, amp=0.08, seed=0, hscale=0.1, vscale=0.005):
    """Low-passed random relief plus a staircase strip: slopes, creases and 0.1 m risers under the robots."""
    rng = np.random.default_rng(seed)
    z = rng.uniform(-1, 1, (rows // 4 + 2, cols // 4 + 2))
    z = np.kron(z, np.ones((4, 4)))[:rows, :cols]
    for _ in range(3):
        z = 0.25 * (np.roll(z, 1, 0) + np.roll(z, -1, 0) + np.roll(z, 1, 1) + np.roll(z, -1, 1))
    z = amp * z / np.abs(z).max()
    z[:, 100:140] += 0.1 * (np.arange(40) // 4)[None, :] % 0.5
    return np.rint(z / vscale).astype(np.int16), hscale, vscale


def scatter_on_field(S, B, g, hs, hscale, vscale, clearance):
    N = B.root_states.shape[1]
    B.root_states[0].uniform_(3.0, 20.0, generator=g)
    B.root_states[1].uniform_(3.0, 20.0, generator=g)
    ix = (B.root_states[0] / hscale).long()
    iy = (B.root_states[1] / hscale).long()
    ground = torch.from_numpy(hs.astype(np.float32))[ix, iy] * vscale
    B.root_states[2] = ground + clearance


@pytest.mark.parametrize("walls", [False, True])
@pytest.mark.parametrize("scenario", ["standing", "tumbling"])
def test_physics_substep_on_height_field(scenario, walls):
    """Same comparison as above on a rough int16 height field with a staircase strip (BASELINE config 3): bilinear height +
    tilted contact frames + world-impulse warm start; walls: the same field as a `trimesh` terrain (slope_treshold 0.75): the
    0.1 m risers of the strip are vertical faces with horizontal contact normals (the kernel's WALLS instance)."""
    N = 256
    cfg, S, meta, Bc, orc = gpu_pair("train", N, extra={"domain_rand": dict(randomize_gravity=False)})
    hs, hscale, vscale = rough_field()
    H.bind_height_field(S, Bc, hs, hscale, vscale, 0.0, slope_threshold=0.75 if walls else None)
    assert S.terrain_type == 1 and (S.hf_wall_units > 0) == walls
    g = torch.Generator().manual_seed(4)
    if scenario == "standing":
        standing_state(S, Bc, z=0.28)
        scatter_on_field(S, Bc, g, hs, hscale, vscale, 0.29)
    else:
        q = torch.randn(4, N, generator=g)
        Bc.root_states[3:7] = q / q.norm(dim=0, keepdim=True)
        scatter_on_field(S, Bc, g, hs, hscale, vscale, 0.0)
        Bc.root_states[2] += torch.empty(N).uniform_(0.08, 0.35, generator=g)
        Bc.root_states[7:13].uniform_(-2, 2, generator=g)
        Bc.dof_vel.uniform_(-5, 5, generator=g)
    if walls:                                                          # half of the robots over the staircase strip
        Bc.root_states[1, ::2].uniform_(10.2, 13.8, generator=g)
        ix, iy = (Bc.root_states[0] / hscale).long(), (Bc.root_states[1] / hscale).long()
        ground = torch.from_numpy(hs.astype(np.float32))[ix, iy] * vscale
        Bc.root_states[2, ::2] = ground[::2] + (0.29 if scenario == "standing" else torch.empty(N // 2).uniform_(0.08, 0.35, generator=g))
    Bc.torques.uniform_(-20, 20, generator=g)
    orc = __import__("pyoracle").Oracle(S, Bc)
    Bg, sim = to_gpu(S, Bc)
    sh = Shadow32(S, Bc, orc)
    att = Attribution(N)
    wall_contacts = 0
    for it in range(8):
        orc.physics_substep()
        sh.o.physics_substep()
        sim.physics_substep()
        torch.cuda.synchronize()
        assert torch.isfinite(Bg.root_states).all() and torch.isfinite(Bg.dof_vel).all()
        att.step(make_ratio(att, SUBSTEP_TOL), Bg, Bc, sh.B)
        wall_contacts += int((Bc.contact_signature[1] & 0x1FFF != 0).sum())
        sync_from(Bc, Bg, sim, orc)
        sh.sync()
    att.finish(f"height-field substep[{scenario}, walls={walls}]")
    cf = Bc.contact_forces.view(17, 3, N)
    assert float(cf[:, 2].abs().max()) > 1.0 and float(cf[:, :2].abs().max()) > 0.5       # tilted normals / friction at work
    assert wall_contacts == 0 if not walls else (wall_contacts > 0 or scenario == "standing"), wall_contacts       # the vertical faces were hit


def run_height_field_comparison(walls, N=256, steps=40, product=False, residual=(0, 0.0), alt=None):
    """full steps on the rough int16 height field of rough_field(): 187-point height scan in the observation, resets onto the
    field, the height-relative termination test (legged_robot.py:160-178, 1793-1806); walls: as a `trimesh` terrain (vertical
    risers).  product: the instance the product launches (no signature code) beside its `_sig` twin."""
    pts_x = [round(-0.8 + 0.1 * i, 1) for i in range(17)]
    pts_y = [round(-0.5 + 0.1 * i, 1) for i in range(11)]
    ex = {"terrain": dict(measure_heights=True, measured_points_x=pts_x, measured_points_y=pts_y),
          "env": dict(observe_heights=True, num_observations=70 + 187),
          "domain_rand": dict(randomize_gravity=False)}
    import pyoracle
    alt = (PARITY_ALT if alt is None else alt) and product
    cfg, S, meta, Bc = make_sim("train_noise", N, seed=1013 if alt else 13, extra=ex)
    hs, hscale, vscale = rough_field(seed=1002 if alt else 2)
    H.bind_height_field(S, Bc, hs, hscale, vscale, 0.0, slope_threshold=0.75 if walls else None)
    randomize_dr(Bc, 13)
    Bc.enable_contact_signature()
    Bc.env_origins[0].uniform_(4.0, 19.0, generator=torch.Generator().manual_seed(1))
    Bc.env_origins[1].uniform_(4.0, 19.0, generator=torch.Generator().manual_seed(2))
    ix = (Bc.env_origins[0] / hscale).long()
    iy = (Bc.env_origins[1] / hscale).long()
    Bc.env_origins[2] = torch.from_numpy(hs.astype(np.float32))[ix, iy] * vscale + 0.05
    orc = pyoracle.Oracle(S, Bc)
    orc.reset_idx()
    # (root 3e-3 here: on the relief the contact normals are the bilinear interpolant's gradient AT the contact point, so a
    #  round-off sized shift of the point tilts the whole contact frame — the flat-terrain tests keep 1e-3)
    # (torques 2e-2: the actuator network's gain on the PREVIOUS substep's position error is 17 N m/rad on average, 23 at the
    #  99 % quantile (finite differences of oracle/pyoracle.actuator_net), so a q inside its 1e-3 tolerance already moves the
    #  torque by 2e-2 N m; the flat-terrain tests keep 5e-3 because q agrees to 1e-4 there)
    keys = (("root_states", 3e-3, 1e-3), ("dof_pos", 1e-3, 0), ("dof_vel", 2e-2, 1e-3), ("rew_buf", 2e-4, 1e-3),
            ("torques", 2e-2, 1e-3), ("foot_positions", 1e-3, 0), ("measured_heights", 1e-3, 0))
    pp = ProductPair(S, Bc, orc, [k for k, _, _ in keys] + ["obs_buf", "obs_history"]) if product else None
    Bg, sim = (pp.Bg, pp.sim) if product else to_gpu(S, Bc)
    pp.sync() if product else sync_from(Bc, Bg, sim, orc)
    sh = Shadow32(S, Bc, orc)
    sp = ShadowPert(S, Bc, orc)
    rng = np.random.default_rng(1000 if alt else 0)
    resets = 0
    att = Attribution(N, residual)
    for step in range(steps):
        a = (rng.standard_normal((N, 12)) * (1.0 if step % 2 else 0.3)).astype(np.float32)
        orc.step(a)
        sh.o.step(a)
        sp.step(a)
        twin = None
        if product:
            twin = pp.step(torch.from_numpy(a).cuda())
        else:
            sim.step(torch.from_numpy(a).cuda())
            torch.cuda.synchronize()
        cpu_reset = Bc.reset_buf.bool()
        # A scan point within round-off of a cell boundary reads the neighbouring sample in fp32 (legged_robot.py:1793-1806 floors
        # (x + border) / scale): environments whose ONLY differences are a few of the 187 scan heights (and their observation
        # columns) are attributed to that — the rest of their state, rewards and the 70 proprioceptive columns must agree
        scan_pts = ((Bg.measured_heights.cpu() - Bc.measured_heights).abs() > 1e-3).sum(0)
        core = make_ratio(att, keys[:-1])(Bg, Bc)
        prop = att.ratio(Bg.obs_buf[:, :70].contiguous(), Bc.obs_buf[:, :70].contiguous(), 5e-3, 1e-3, env_dim=0)
        scan_flip = (scan_pts > 0) & (scan_pts <= 4) & (core <= 1.0) & (prop <= 1.0)
        att.step(make_ratio(att, keys, (("obs_buf", 5e-3, 1e-3),)), Bg, Bc, sh.B, reset_key="reset_buf", also_attributed=scan_flip, twin=twin, pert=sp)
        resets += int(cpu_reset.sum())
        pp.sync() if product else sync_from(Bc, Bg, sim, orc)
        sh.sync()
        sp.sync()
    assert int(Bg.fault_counts[:10].sum()) == 0, Bg.fault_counts.tolist()
    att.note = pp.note() if product else ""
    att.finish(f"height-field full step (walls={walls}{', PRODUCT instance' if product else ''}{', alt seed + relief' if alt else ''}) [{N} envs x {steps} steps]")
    assert Bc.obs_buf.shape[1] == 257 and float(Bc.obs_buf[:, 70:].abs().max()) > 0.1
    assert resets > 5 or steps < 40
    return att, pp


@pytest.mark.parametrize("walls", [False, True])
def test_full_step_on_height_field(walls):
    """40 full steps with the 187-point height scan in the observation, resets onto the field and the
    height-relative termination test (legged_robot.py:160-178, 1793-1806); walls: as a `trimesh` terrain (vertical risers)."""
    run_height_field_comparison(walls)


# (instance, environments, steps, alt): the three instances at configs[1] / [2]'s 4096 environments; round 6 (the review of round 5): plane and walls at
# configs[4]'s per-GPU size — 8192 environments = 512 workgroups on 256 CUs: the only regime in which a workgroup starts on a CU whose LDS the
# previous one just left (step_body zero-fills `lds` / `ldsx`) — and the second seed / second relief that used to sit behind GO1_PARITY_ALT=1
PRODUCT_CASES = [("plane", 4096, 40, False), ("hf", 4096, 40, False), ("walls", 4096, 40, False),
                 ("plane", 8192, 10, False), ("hf", 8192, 10, False), ("walls", 8192, 10, False),
                 ("plane", 4096, 40, True), ("hf", 4096, 40, True), ("walls", 4096, 40, True)]


@pytest.mark.parametrize("instance,envs,steps,alt", PRODUCT_CASES,
                         ids=[f"{i}{'' if n == 4096 else f'-{n}'}{'-alt' if a else ''}" for i, n, _, a in PRODUCT_CASES])
def test_product_instances_match_oracle(instance, envs, steps, alt):
    """The kernels the PRODUCT launches and bench.py times — go1_step_kernel (plane), go1_step_kernel_hf, go1_step_kernel_walls:
    the template instances WITHOUT the contact-signature code (csrc/go1sim.hip `launch`: chosen when Go1SimBuffers.contact_signature
    is NULL) — against the oracle at BASELINE configs[1] / [2]'s 4096 environments, 40 full steps re-synchronised every step
    (reference legged_robot.py:60-88, 907-946), tolerances of FULL_STEP_TOL (relief: those of test_full_step_on_height_field).
    Attribution by the fp32-oracle rules (b) / (c); rule (a) only for an environment-step whose outputs are bit-identical in
    the product instance and in its `_sig` twin stepped beside it (module docstring) — the count of environment-steps in which
    the two instances differ at all is part of the summary line.

    At this size — 16 x the environment-steps of test_full_step_on_height_field — the relief runs meet states the 256-environment runs never
    did: twice in 163,840 environment-steps the hardware kernel left a tolerance by x1.8 with identical contact lists and per-sweep active
    sets while the fp32 oracle stayed at 0.06-0.09 of it.  Replayed (tools/debug/hf_env_replay.py, hf_env_substeps.py: deterministic, the same
    for one environment alone and in a full wavefront, unchanged by -O1 / -ffp-contract=off) both turned out to be ILL-CONDITIONED states — a
    leg-leg contact of 85 N appearing in the step's last substep, an impact on a calf — in which the fp64 oracle's OWN result moves by
    0.011-0.08 rad/s when its inputs are perturbed by one fp32 ulp (ShadowPert above; 2e-5 on an ordinary state): the fp32 oracle's small
    error there was one lucky draw.  Rule (b) therefore takes the conditioning from the larger of the fp32 oracle's error and of that
    perturbation probe; with it every environment-step of the three runs is attributed (no residual category)."""
    N = envs
    if os.environ.get("GO1_PRODUCT_PARITY_ENVS"):         # (tools/dry_run_gpu_tests.py: the emulator needs a smaller count; the 8192 cases keep their ratio)
        N = int(os.environ["GO1_PRODUCT_PARITY_ENVS"]) * envs // 4096
        steps = steps if N >= 4096 else 6
    if instance == "plane":
        att, resets, resamples, _ = run_full_step_comparison("train_noise", N, steps, what="PRODUCT instance, plane", product=True, alt=alt)
        assert (resets > N // 64 and resamples > N // 64) or steps < 40
    else:
        att, pp = run_height_field_comparison(instance == "walls", N=N, steps=steps, product=True, alt=alt)
        # the twin relation is asserted, not only reported, since round 5 met a build in which it did not hold (1 % of the `_hf` instance's
        # environment-steps: fused multiply-adds formed differently in the two instances — csrc/go1_physics.h GO1_NO_CONTRACT,
        # tests/twin_probe.py tells a compile difference from a race)
        assert pp.differ == 0, pp.note()
    assert att.env_steps == N * steps


def test_determinism_and_shard_independence():
    """Same seed -> bit-identical results; envs [256,512) of a 512-env run == a 256-env run with env_id_offset 256."""
    N = 512
    cfg, S, meta, Bc, orc = gpu_pair("train_noise", N, seed=5)
    acts = torch.randn(6, N, 12)

    def run(Ssub, Bsub, a):
        Bg = Bsub.clone_to("cuda:0")
        sim = H.Go1Sim(Ssub, Bg, 0)
        for t in range(a.shape[0]):
            sim.step(a[t].cuda().contiguous())
        torch.cuda.synchronize()
        return Bg
    B1 = run(S, Bc, acts)
    B2 = run(S, Bc, acts)
    for k in ("root_states", "dof_pos", "obs_buf", "rew_buf", "commands"):
        assert torch.equal(B1.tensors[k], B2.tensors[k]), k
    cfg2, S2, meta2, Bs = make_sim("train_noise", 256, seed=5, env_id_offset=256)
    for k, t in Bc.tensors.items():
        if t is None or k not in Bs.tensors or Bs.tensors[k] is None:
            continue
        if t.shape == Bs.tensors[k].shape:
            Bs.tensors[k].copy_(t)
        elif t.shape[-1] == N and t.dim() >= 1 and Bs.tensors[k].shape[-1] == 256:
            Bs.tensors[k].copy_(t[..., 256:])
        elif t.shape[0] == N:
            Bs.tensors[k].copy_(t[256:])
    B3 = run(S2, Bs, acts[:, 256:])
    for k in ("root_states", "dof_pos", "rew_buf", "commands"):
        assert torch.equal(B3.tensors[k], B1.tensors[k][..., 256:]), k
    assert torch.equal(B3.obs_buf, B1.obs_buf[256:])


def test_full_size_invariants_4096():
    """BASELINE config 2 size: 4096 envs, 50 steps of N(0,1) actions: finite, limits, cone, weight support at rest."""
    N = 4096
    cfg, S, meta, Bc, orc = gpu_pair("train_noise", N, seed=9, extra={"domain_rand": dict(randomize_gravity=False)})
    Bg, sim = to_gpu(S, Bc)
    g = torch.Generator(device="cuda").manual_seed(0)
    for t in range(50):
        a = torch.randn(N, 12, device="cuda", generator=g)
        sim.step(a)
    torch.cuda.synchronize()
    for k in ("root_states", "dof_pos", "dof_vel", "obs_buf", "rew_buf", "contact_forces", "obs_history"):
        assert torch.isfinite(Bg.tensors[k]).all(), k
    assert float(Bg.torques.abs().max()) <= 33.5 + 1e-4
    lo = torch.tensor([-0.802851455917, -1.0471975512, -2.69653369433] * 4, device="cuda").unsqueeze(1)
    hi = torch.tensor([0.802851455917, 4.18879020479, -0.916297857297] * 4, device="cuda").unsqueeze(1)
    live = Bg.reset_buf == 0
    assert bool(((Bg.dof_pos >= lo - 0.03) & (Bg.dof_pos <= hi + 0.03))[:, live].all())      # limit rows: solver residual, not a clamp
    cf = Bg.contact_forces.view(17, 3, N)
    mu = 0.5 * (Bg.friction_coeffs + 1.0)
    ft = torch.sqrt(cf[:, 0] ** 2 + cf[:, 1] ** 2)
    assert bool((ft <= mu * cf[:, 2] * (1 + 1e-3) + 1e-3).all())
    assert bool((cf[:, 2] >= 0).all())
    # then stand still: total normal force equals the weight
    z = torch.zeros(N, 12, device="cuda")
    for t in range(150):
        sim.step(z)
    torch.cuda.synchronize()
    cf = Bg.contact_forces.view(17, 3, N)
    weight = (11.309932 + Bg.payloads) * 9.8
    settled = (Bg.episode_length_buf > 100) & (Bg.root_states[7:13].abs().max(0).values < 0.05)
    assert int(settled.sum()) > N // 4
    err = ((cf[:, 2].sum(0) - weight).abs() / weight)[settled]
    assert float(err.median()) < 0.02
    # history ring: the reference window equals the last H observations, newest last
    H_, no = S.num_obs_history, S.num_obs
    c, _ = sim.counters()
    off = sim.history_window_offset()
    assert off == ((c % (H_ + 1)) + 1) % (H_ + 1) * no
    win = Bg.obs_history[:, off:off + H_ * no]
    assert torch.equal(win[:, -no:], Bg.obs_buf)


def test_failed_simulation_guard():
    """A non-finite reward input (here: a NaN in the previous joint velocity of one environment, which only the dof_acc
    term reads) must not leave the kernel: the term counts as 0, the episode ends, every other environment and the
    running sums stay untouched and finite."""
    N = 64
    cfg, S, meta, Bc, orc = gpu_pair("train_noise", N, seed=21)
    Bg, sim = to_gpu(S, Bc)
    Bref, sim_ref = to_gpu(S, Bc)
    sync_from(Bc, Bg, sim, orc)
    sync_from(Bc, Bref, sim_ref, orc)
    victim = 37
    Bg.last_dof_vel[4, victim] = float("nan")
    a = torch.zeros(N, 12, device="cuda")
    sim.step(a)
    sim_ref.step(a)
    torch.cuda.synchronize()
    assert torch.isfinite(Bg.rew_buf).all() and torch.isfinite(Bg.episode_sums).all() and torch.isfinite(Bg.obs_buf).all()
    assert int(Bg.reset_buf[victim]) == 1 and int(Bg.time_out_buf[victim]) == 0
    assert int(Bg.episode_length_buf[victim]) == 0                                   # re-initialised in the same step
    others = torch.arange(N, device="cuda") != victim
    torch.testing.assert_close(Bg.rew_buf[others], Bref.rew_buf[others], rtol=0, atol=0)
    assert torch.equal(Bg.reset_buf[others], Bref.reset_buf[others])
    # the containment is reported, not hidden: fault word of the victim, one count, nothing anywhere else
    bit = 1 << H.abi.GO1_FAULT_REWARD
    assert int(Bg.fault_flags[victim]) & H.FAULT_FATAL_MASK == bit and int((Bg.fault_flags[others] & H.FAULT_FATAL_MASK).sum()) == 0
    assert Bg.fault_counts.tolist()[H.abi.GO1_FAULT_REWARD] >= 1 and int(Bref.fault_counts[:10].sum()) == 0


@pytest.mark.parametrize("where", ["root_z", "quat", "dof_vel", "base_ang_vel"])
def test_failed_state_is_contained_and_reported(where):
    """A non-finite STATE (injected; PhysX never produces one, legged_robot.py:76-80) is caught at the site, the
    environment is re-initialised in the same launch and no buffer the policy or the logger reads keeps a non-finite
    value; all other environments are bit-identical to an undisturbed run."""
    N = 128
    cfg, S, meta, Bc, orc = gpu_pair("train_noise", N, seed=23)
    Bg, sim = to_gpu(S, Bc)
    Bref, sim_ref = to_gpu(S, Bc)
    for _ in range(3):
        a = torch.zeros(N, 12, device="cuda")
        sim.step(a); sim_ref.step(a)
    victim = 77
    {"root_z": lambda: Bg.root_states[2].__setitem__(victim, float("nan")),
     "quat": lambda: Bg.root_states[4].__setitem__(victim, float("inf")),
     "dof_vel": lambda: Bg.dof_vel[7].__setitem__(victim, float("nan")),
     "base_ang_vel": lambda: Bg.root_states[11].__setitem__(victim, float("-inf"))}[where]()
    for _ in range(3):
        a = torch.zeros(N, 12, device="cuda")
        sim.step(a); sim_ref.step(a)
    torch.cuda.synchronize()
    for k, t in Bg.tensors.items():
        if t is not None and t.is_floating_point() and k != "episode_log":
            assert torch.isfinite(t).all(), k
    word = int(Bg.fault_flags[victim])
    assert word & (1 << H.abi.GO1_FAULT_STATE_IN) and word & H.FAULT_FATAL_MASK
    others = torch.arange(N, device="cuda") != victim
    assert int((Bg.fault_flags[others] & H.FAULT_FATAL_MASK).sum()) == 0 and int(Bref.fault_counts[:10].sum()) == 0
    for k in ("root_states", "dof_pos", "obs_buf", "rew_buf"):
        a_, b_ = Bg.tensors[k], Bref.tensors[k]
        if a_.shape[-1] == N:
            assert torch.equal(a_[..., others], b_[..., others]), k
        else:
            assert torch.equal(a_[others], b_[others]), k


def test_train_eval_split_matches_oracle():
    """eval_cfg (reference base_task.py:43-49, legged_robot.py:531-544): 256 training + 128 evaluation environments, the
    evaluation group with its own domain-randomisation ranges, pushes and reset distribution (go1sim_set_eval_config:
    second configuration block selected per wavefront).  HIP kernel vs oracle on identical streams through 40 steps full of
    time-outs; every re-drawn parameter lies in ITS group's range; evaluation episodes stay out of the training log."""
    import pyoracle
    N, NT = 384, 256
    ev = {"domain_rand": dict(friction_range=[5.0, 5.5], restitution_range=[0.7, 0.8], added_mass_range=[4.0, 4.5],
                              motor_strength_range=[1.5, 1.6], motor_offset_range=[0.10, 0.11], push_robots=True, max_push_vel_xy=2.0,
                              randomize_rigids_after_start=True, randomize_friction=True, randomize_restitution=True, randomize_base_mass=True),
          "terrain": dict(yaw_init_range=0.1)}
    cfg, S, meta, Bc = make_sim("dr", N, seed=5)
    _, S_ev_full, _, _ = make_sim("dr", N, seed=5, extra=ev)
    S_eval = H.make_eval_sim_config(S, S_ev_full)
    randomize_dr(Bc, 5)
    Bc.enable_contact_signature()
    Bc.episode_sums_eval.fill_(-1.0)
    orc = pyoracle.Oracle(S, Bc)
    orc.set_eval_config(S_eval, NT)
    orc.reset_idx()
    Bc.episode_length_buf[:] = torch.randint(int(S.max_episode_length) - 30, int(S.max_episode_length) - 1, (N,), dtype=torch.int32,
                                             generator=torch.Generator().manual_seed(1))
    Bg, sim = to_gpu(S, Bc)
    sim.set_eval_config(S_eval, NT)
    with pytest.raises(RuntimeError):
        sim.set_eval_config(S_eval, 250)                    # not a multiple of 16
    sync_from(Bc, Bg, sim, orc)
    sh = Shadow32(S, Bc, orc)
    sh.o.set_eval_config(S_eval, NT)
    rng = np.random.default_rng(2)
    train_resets = 0
    att = Attribution(N)          # no free budget: an environment outside the tolerances must be attributed (module docstring)
    keys = (("root_states", 1e-3, 1e-3), ("dof_pos", 1e-3, 0), ("friction_coeffs", 1e-6, 0), ("restitutions", 1e-6, 0),
            ("payloads", 1e-6, 0), ("motor_strengths", 1e-6, 0), ("motor_offsets", 1e-6, 0), ("rew_buf", 2e-4, 1e-3),
            ("commands", 1e-5, 0), ("episode_sums", 1e-3, 1e-3), ("episode_sums_eval", 1e-3, 1e-3))
    for step in range(40):
        a = (rng.standard_normal((N, 12)) * 0.5).astype(np.float32)
        Bg.episode_log.zero_()
        orc.step(a)
        sh.o.step(a)
        sim.step(torch.from_numpy(a).cuda())
        torch.cuda.synchronize()
        bad_env = att.step(make_ratio(att, keys), Bg, Bc, sh.B, reset_key="reset_buf", also_attributed=grazing_collision_count(att, Bg, Bc, keys))
        good = ~bad_env
        n_tr = int((Bc.reset_buf[:NT].bool() & good[:NT]).sum())
        if bool(good.all()):
            np.testing.assert_allclose(Bg.episode_log.cpu().numpy(), Bc.episode_log.numpy(), rtol=1e-3, atol=1e-3)
            assert int(round(float(Bg.episode_log[-1]))) == int(Bc.reset_buf[:NT].sum())        # training resets only
        train_resets += n_tr
        sync_from(Bc, Bg, sim, orc)
        sh.sync()
    att.finish("train / evaluation split [dr, 384 envs x 40 steps]")
    done = Bg.episode_sums_eval[-1].cpu() != -1.0
    assert int(done[NT:].sum()) > 50 and int(done[:NT].sum()) == 0 and train_resets > 50
    fr, ms, pl = Bg.friction_coeffs.cpu(), Bg.motor_strengths.cpu(), Bg.payloads.cpu()
    assert bool(((fr[NT:] >= 5.0) & (fr[NT:] <= 5.5)).all()) and bool((fr[:NT] < 5.0).all())
    assert bool(((ms[:, NT:] >= 1.5) & (ms[:, NT:] <= 1.6)).all()) and bool((ms[:, :NT] < 1.5).all())
    assert bool(((pl[NT:] >= 4.0) & (pl[NT:] <= 4.5)).all())
