"""The PPO update ON THE GPU against the reference's own numbers: GPU twins of tests/test_ppo.py::test_update_matches_reference.

tests/golden/ppo.npz and ppo_fuzz*.npz hold a rollout pushed through the REFERENCE go1_gym_learn.ppo_cse (RolloutStorage.compute_returns +
PPO.update, fp32, CPU; tests/golden/make_golden.py executes the reference's classes).  tests/test_ppo.py pins this repository's fp32 path to them
on the CPU; here the same path runs on cuda:0 (fp32 GEMMs of rocBLAS / hipBLASLt, the device-side adaptive learning rate, the flat gradient
buffer) and must meet the SAME tolerances.  The bf16 fused update (csrc/go1ppo.hip) is compared with this fp32 path in
tests/test_gpu_ppo_fused.py: reference == fp32 CPU == fp32 GPU (here, same fixtures, same tolerances) ~ bf16 fused (there, bf16 tolerances).
Reference: go1_gym_learn/ppo_cse/ppo.py:97-205, rollout_storage.py:76-139."""
import json
import os

import numpy as np
import pytest
import torch

from test_ppo import ppo_args_guard, small_ac_args  # noqa: F401  (fixtures)
from util import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fname", ["ppo.npz", "ppo_fuzz0.npz", "ppo_fuzz1.npz", "ppo_fuzz2.npz", "ppo_fuzz3.npz"])
def test_gpu_fp32_update_matches_reference(small_ac_args, ppo_args_guard, fname, monkeypatch):  # noqa: F811
    from go1_gym_learn.ppo_cse.actor_critic import ActorCritic
    from go1_gym_learn.ppo_cse.ppo import PPO, PPO_Args
    monkeypatch.delenv("GO1_POLICY_DTYPE", raising=False)
    d = np.load(os.path.join(GOLDEN, fname))
    for k, v in (json.loads(str(d["ppo_args"])) if "ppo_args" in d.files else {}).items():
        setattr(PPO_Args, k, v)
    PPO_Args.autocast_bf16 = False
    N, T, no, npv, H, na = [int(x) for x in d["dims"]]
    ac = ActorCritic(no, npv, no * H, na)
    ac.load_state_dict({k[5:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("init_")})
    alg = PPO(ac, device="cuda:0")
    assert not alg.bf16 and not alg.fused
    alg.init_storage(N, T, [no], [npv], [no * H], [na])
    st = alg.storage
    dev = lambda a: torch.from_numpy(a).to("cuda:0")
    for k in ("observations", "privileged_observations", "actions", "rewards", "dones", "values", "mu", "sigma", "actions_log_prob"):
        getattr(st, k).copy_(dev(d["in_" + k]))
    for t in range(T):
        st.write_history(st.observation_histories[t], dev(d["in_observation_histories"][t]), dev(d["in_privileged_observations"][t]))
    st.step = T
    st.compute_returns(dev(d["last_values"]), PPO_Args.gamma, PPO_Args.lam)
    np.testing.assert_allclose(st.returns.cpu().numpy(), d["out_returns"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(st.advantages.cpu().numpy(), d["out_advantages"], rtol=1e-5, atol=1e-5)
    # the reference drew its mini-batch permutations from the CPU generator (torch.randperm on the storage's device = cpu there); on cuda:0
    # torch.randperm reads the device generator — another stream for the same seed.  The twin takes the CPU draws and moves them over.
    cpu_randperm = torch.randperm
    monkeypatch.setattr(torch, "randperm", lambda n, **kw: cpu_randperm(n).to(kw.get("device", "cpu")))
    torch.manual_seed(int(d["seed"]) + 2)
    losses = alg.update()
    np.testing.assert_allclose(losses, d["losses"], rtol=2e-4, atol=1e-6)
    assert alg.learning_rate == pytest.approx(float(d["final_lr"]), rel=1e-6)
    for k, v in alg.sync_module().state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), d["final_" + k], rtol=2e-3, atol=2e-5, err_msg=k)
