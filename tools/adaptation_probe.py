"""Sim-to-sim indicator (optional probe of the round-2 review, item 6): the reference's PRETRAINED adaptation module — trained on Isaac
Gym rollouts of the reference's own policy to recover (friction, restitution) from 30-step observation histories — is fed the
observation histories THIS simulator produces under a policy trained HERE with the same configuration, and its predictions are
compared with the true privileged observations (same normalisation: tests/golden maps fixtures), next to the adaptation module
trained here and to the constant predictor.

Limits, stated: the reference module saw the gait of ITS policy; the histories here come from another policy (same rewards and
command distribution, different weights), so a miss is inconclusive; a hit means the contact / friction signatures in the
observations (joint velocities and positions under slip, action history) are close enough to PhysX's for a network that never
saw this simulator.  Weights: tests/golden/ref_adaptation_module.npz (written by tests/golden/make_golden.py
ref_adaptation_module where /root/reference exists).

    python tools/adaptation_probe.py [--iters 2500]        (GPU box; summary to stdout -> profiles/r03_adaptation_probe.txt)"""
import argparse
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "walk-these-ways_amd")
for p in (os.path.join(PKG, "shims"), PKG, REPO, os.path.join(REPO, "tools")):
    sys.path.insert(0, p)
import torch  # noqa: E402


def reference_module(device):
    d = np.load(os.path.join(REPO, "tests", "golden", "ref_adaptation_module.npz"))
    net = torch.nn.Sequential(torch.nn.Linear(2100, 256), torch.nn.ELU(), torch.nn.Linear(256, 128), torch.nn.ELU(), torch.nn.Linear(128, 2))
    with torch.no_grad():
        for i in (0, 2, 4):
            net[i].weight.copy_(torch.from_numpy(d[f"w{i}_weight"].astype(np.float32)))
            net[i].bias.copy_(torch.from_numpy(d[f"w{i}_bias"].astype(np.float32)))
        err = float((net(torch.from_numpy(d["probe_in"])) - torch.from_numpy(d["probe_out"])).abs().max())
    assert err < 2e-3, err          # (CPU BLAS builds differ in the last bits of a 2100-term dot product) against the TorchScript module's own output
    return net.to(device).eval(), err


def stats(pred, true):
    out = []
    for k in range(true.shape[1]):
        p, t = pred[:, k].double(), true[:, k].double()
        r = float(torch.corrcoef(torch.stack((p, t)))[0, 1])
        out.append(dict(r=r, rmse=float((p - t).pow(2).mean().sqrt()), const_rmse=float(t.std()), bias=float((p - t).mean())))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=2500)
    ap.add_argument("--steps", type=int, default=300)
    args = ap.parse_args()
    import play_eval
    say = lambda m: print(m, flush=True)

    def after(runner, env, obs_dict):
        alg = runner.alg
        policy = alg.sync_module()
        policy.eval()
        ref, err = reference_module(env.env.device)
        say(f"# reference adaptation module rebuilt from its weights: max |out - TorchScript out| {err:.1e} on its probe inputs")
        H, P, own = [], [], []
        with torch.inference_mode():
            obs = env.get_observations()
            for i in range(args.steps):
                actions = policy.act_student(obs["obs_history"])
                obs, rew, done, info = env.step(actions)
                if i >= 60 and i % 10 == 0:
                    H.append(obs["obs_history"].float().clone()); P.append(obs["privileged_obs"].float().clone())
            Hc, Pc = torch.cat(H), torch.cat(P)
            pr = ref(Hc)
            po = policy.adaptation_module(Hc)
        names = ("friction", "restitution")
        say(f"# {Hc.shape[0]} (environment, time) samples: deterministic student policy on the training environments after {args.iters} iterations; "
            f"targets = privileged observations (normalised: friction (mu - 0.5) * 2 over mu in [0.1, 3], restitution (e - 0.5) * 2 over e in [0, 0.4])")
        for label, pred in (("reference module (trained on Isaac Gym)", pr), ("module trained here", po)):
            for nm, s in zip(names, stats(pred, Pc)):
                say(f"{label:42s} {nm:12s} Pearson r {s['r']:+.3f}  RMSE {s['rmse']:.3f}  (constant predictor {s['const_rmse']:.3f})  mean error {s['bias']:+.3f}")
        # the friction signal lives at the slippery end: correlation over the samples with mu < 1
        lo = Pc[:, 0] < 1.0
        if int(lo.sum()) > 100:
            for label, pred in (("reference module, mu < 1 only", pr), ("module trained here, mu < 1 only", po)):
                s = stats(pred[lo], Pc[lo])[0]
                say(f"{label:42s} {'friction':12s} Pearson r {s['r']:+.3f}  RMSE {s['rmse']:.3f}  (constant predictor {s['const_rmse']:.3f})")

    play_eval.train_and_evaluate(args.iters, eval_at=[], vxs=(), log_every=500, out=say, after=after)


if __name__ == "__main__":
    main()
