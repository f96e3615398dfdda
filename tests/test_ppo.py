"""PPO side (PyTorch-ROCm) vs the reference classes: the golden rollout in tests/golden/ppo.npz was pushed through
the REFERENCE go1_gym_learn.ppo_cse (RolloutStorage.compute_returns + PPO.update, fp32, CPU) by make_golden.py;
our restructured implementation (fused first layer, flat gradient buffer, device-side adaptive LR, explicit
Gaussian algebra) must reproduce returns, advantages, losses, final learning rate and final weights."""
import os

import numpy as np
import pytest
import torch

from util import GOLDEN


@pytest.fixture()
def small_ac_args():
    from go1_gym_learn.ppo_cse.actor_critic import AC_Args
    old = (AC_Args.actor_hidden_dims, AC_Args.critic_hidden_dims, AC_Args.adaptation_module_branch_hidden_dims)
    AC_Args.actor_hidden_dims, AC_Args.critic_hidden_dims, AC_Args.adaptation_module_branch_hidden_dims = [32, 16], [24, 16], [16, 8]
    yield
    AC_Args.actor_hidden_dims, AC_Args.critic_hidden_dims, AC_Args.adaptation_module_branch_hidden_dims = old


@pytest.fixture()
def ppo_args_guard():
    from go1_gym_learn.ppo_cse.ppo import PPO_Args
    saved = {k: getattr(PPO_Args, k) for k in dir(PPO_Args) if not k.startswith("_") and not callable(getattr(PPO_Args, k))}
    yield PPO_Args
    for k, v in saved.items():
        setattr(PPO_Args, k, v)


@pytest.mark.parametrize("fname", ["ppo.npz", "ppo_fuzz0.npz", "ppo_fuzz1.npz", "ppo_fuzz2.npz", "ppo_fuzz3.npz"])
def test_update_matches_reference(small_ac_args, ppo_args_guard, fname):
    """ppo.npz: train.py's PPO_Args; ppo_fuzz*.npz: four other settings (fixed schedule, plain value loss, several adaptation
    sub-steps, selective adaptation loss, other coefficients / epochs / batch counts; make_golden.py PPO_FUZZ)."""
    import json
    from go1_gym_learn.ppo_cse.actor_critic import ActorCritic
    from go1_gym_learn.ppo_cse.ppo import PPO, PPO_Args
    d = np.load(os.path.join(GOLDEN, fname))
    for k, v in (json.loads(str(d["ppo_args"])) if "ppo_args" in d.files else {}).items():
        setattr(PPO_Args, k, v)
    N, T, no, npv, H, na = [int(x) for x in d["dims"]]
    ac = ActorCritic(no, npv, no * H, na)
    ac.load_state_dict({k[5:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("init_")})
    alg = PPO(ac, device="cpu")
    alg.init_storage(N, T, [no], [npv], [no * H], [na])
    st = alg.storage
    for k in ("observations", "privileged_observations", "actions", "rewards", "dones", "values", "mu", "sigma", "actions_log_prob"):
        getattr(st, k).copy_(torch.from_numpy(d["in_" + k]))
    for t in range(T):       # our storage keeps the history rows augmented: [h, 1, privileged, 0]
        st.write_history(st.observation_histories[t], torch.from_numpy(d["in_observation_histories"][t]),
                         torch.from_numpy(d["in_privileged_observations"][t]))
    st.step = T
    st.compute_returns(torch.from_numpy(d["last_values"]), PPO_Args.gamma, PPO_Args.lam)
    np.testing.assert_allclose(st.returns.numpy(), d["out_returns"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(st.advantages.numpy(), d["out_advantages"], rtol=1e-5, atol=1e-5)
    torch.manual_seed(int(d["seed"]) + 2)          # same randperm as the reference run
    losses = alg.update()
    np.testing.assert_allclose(losses, d["losses"], rtol=2e-4, atol=1e-6)
    assert alg.learning_rate == pytest.approx(float(d["final_lr"]), rel=1e-6)
    for k, v in alg.sync_module().state_dict().items():
        np.testing.assert_allclose(v.numpy(), d["final_" + k], rtol=2e-3, atol=2e-5, err_msg=k)


def test_fused_forward_equals_reference_surface():
    from go1_gym_learn.ppo_cse.actor_critic import ActorCritic
    from go1_gym_learn.ppo_cse.ppo import gaussian_entropy, gaussian_log_prob
    torch.manual_seed(0)
    ac = ActorCritic(70, 2, 2100, 12).double()
    hist, priv = torch.randn(16, 2100).double(), torch.randn(16, 2).double()
    mean, value, latent = ac.fused_forward(hist, priv)
    ac.update_distribution(hist)
    assert (mean - ac.action_mean).abs().max() < 1e-12
    assert (value - ac.evaluate(hist, priv)).abs().max() < 1e-12
    assert (latent - ac.get_student_latent(hist)).abs().max() < 1e-12
    padded = torch.nn.functional.pad(hist, (0, 4))
    assert (ac.fused_forward(padded, priv)[0] - mean).abs().max() < 1e-12
    assert (ac.latent_padded(padded) - latent).abs().max() < 1e-12
    # augmented layout used with the bf16 storage: [h, 1, privileged, 0] -> biases and critic privileged weights in the GEMM
    aug = torch.cat((hist, torch.ones(16, 1).double(), priv, torch.zeros(16, 1).double()), dim=1)
    m2, v2, l2 = ac.fused_forward(aug, None, augmented=True)
    assert (m2 - mean).abs().max() < 1e-12 and (v2 - value).abs().max() < 1e-12 and (l2 - latent).abs().max() < 1e-12
    assert (ac.latent_padded(aug, augmented=True) - latent).abs().max() < 1e-12
    a = ac.distribution.sample()
    assert (gaussian_log_prob(a, mean, ac.std) - ac.get_actions_log_prob(a)).abs().max() < 1e-10
    assert (gaussian_entropy(ac.std) - ac.entropy).abs().max() < 1e-10
    assert sum(p.numel() for p in ActorCritic(70, 2, 2100, 12).parameters()) == 3054619


def test_state_dict_keys_match_reference_layout():
    from go1_gym_learn.ppo_cse.actor_critic import ActorCritic
    keys = set(ActorCritic(70, 2, 2100, 12).state_dict())
    expect = {"std"} | {f"{m}.{i}.{w}" for m, idx in (("adaptation_module", (0, 2, 4)), ("actor_body", (0, 2, 4, 6)),
                                                         ("critic_body", (0, 2, 4, 6))) for i in idx for w in ("weight", "bias")}
    assert keys == expect
    ac = ActorCritic(70, 2, 2100, 12)
    torch.jit.script(ac.adaptation_module)      # export path of Runner.save
    torch.jit.script(ac.actor_body)


def test_flat_layout_adaptation_prefix_and_roundtrip():
    """FlatPolicy layout: pack/unpack is exact, the adaptation module's parameters (tail blocks + its rows of W1) are
    one contiguous prefix [0, adaptation_numel) — what the adaptation optimiser / gradient all-reduce cover — and
    the padded columns / rows stay exactly zero."""
    from go1_gym_learn.ppo_cse.actor_critic import ActorCritic
    from go1_gym_learn.ppo_cse.flat_policy import FlatPolicy, HEAD_COLS
    torch.manual_seed(0)
    ac = ActorCritic(70, 2, 2100, 12)
    pol = FlatPolicy(ac)
    flat = torch.zeros(pol.numel)
    pol.pack(ac, flat)
    names = [n for n, _ in pol.blocks]
    assert names[:4] == ["adaptation.1.W", "adaptation.1.b", "adaptation.2.W", "adaptation.2.b"] and names[4] == "W1"
    n_ad = sum(p.numel() for p in ac.adaptation_module.parameters())
    prefix = flat[:pol.adaptation_numel]
    assert int((prefix != 0).sum()) == n_ad                       # every adaptation parameter, nothing else
    assert not flat[pol.adaptation_numel:pol.adaptation_numel + 8].eq(0).all()        # actor rows of W1 follow directly
    W1 = pol._block(flat, "W1")
    nd, na = pol.first[0], pol.first[1]
    assert not W1[:nd + na, pol.K + 1:].any()                     # adaptation / actor rows never see the privileged columns
    assert not pol._block(flat, "Wz")[:, pol.npv:].any() and pol._block(flat, "Wz").shape[1] == HEAD_COLS
    ac2 = ActorCritic(70, 2, 2100, 12)
    FlatPolicy(ac2).unpack(flat, ac2)              # a FlatPolicy is bound to the module it was built from
    for (k, a), (_, b) in zip(ac.state_dict().items(), ac2.state_dict().items()):
        if k != "std":
            assert torch.equal(a, b), k
    # functional forward on the flat buffer == the nn.Module
    h, p = torch.randn(5, 2100), torch.randn(5, 2)
    x = torch.zeros(5, pol.Kp)
    x[:, :2100], x[:, 2100], x[:, 2101:2103] = h, 1.0, p
    mean, value, latent = pol.forward(flat, x)
    torch.testing.assert_close(mean, ac.act_inference({"obs_history": h, "privileged_obs": p}), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(value, ac.evaluate(h, p), rtol=1e-4, atol=1e-5)


def test_storage_returns_are_in_place():
    """the update's HIP graphs (and the fused loss kernel) hold the addresses of advantages / returns."""
    from go1_gym_learn.ppo_cse.rollout_storage import RolloutStorage
    st = RolloutStorage(6, 4, [3], [2], [9], [2])
    st.rewards.normal_(); st.values.normal_()
    pa, pr = st.advantages.data_ptr(), st.returns.data_ptr()
    st.compute_returns(torch.randn(6, 1), 0.99, 0.95)
    assert st.advantages.data_ptr() == pa and st.returns.data_ptr() == pr
    assert abs(float(st.advantages.mean())) < 1e-6 and abs(float(st.advantages.std()) - 1.0) < 1e-4


# ---- the older teacher-student runner (go1_gym_learn.ppo) ----------------------------------------
@pytest.fixture()
def small_rma_args():
    from go1_gym_learn.ppo.actor_critic import AC_Args
    keys = ("actor_hidden_dims", "critic_hidden_dims", "adaptation_module_branch_hidden_dims", "env_factor_encoder_branch_input_dims",
            "env_factor_encoder_branch_latent_dims", "env_factor_encoder_branch_hidden_dims")
    old = {k: getattr(AC_Args, k) for k in keys}
    AC_Args.actor_hidden_dims, AC_Args.critic_hidden_dims = [32, 16], [24, 16]
    AC_Args.adaptation_module_branch_hidden_dims = [[16, 8]]
    AC_Args.env_factor_encoder_branch_input_dims, AC_Args.env_factor_encoder_branch_latent_dims = [5], [4]
    AC_Args.env_factor_encoder_branch_hidden_dims = [[12, 8]]
    yield
    for k, v in old.items():
        setattr(AC_Args, k, v)


def test_teacher_student_runner_matches_reference(small_rma_args):
    """tests/golden/ppo_rma.npz: the REFERENCE go1_gym_learn.ppo classes (executed by make_golden.py) on a fixed rollout —
    same state_dict keys, same teacher / student / value outputs, same returns, losses, learning rate and weights
    after PPO.update()."""
    from go1_gym_learn.ppo import ActorCritic
    from go1_gym_learn.ppo.ppo import PPO, PPO_Args
    d = np.load(os.path.join(GOLDEN, "ppo_rma.npz"))
    N, T, no, npv, H, na = [int(x) for x in d["dims"]]
    ac = ActorCritic(no, npv, no * H, na)
    init = {k[5:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("init_")}
    assert sorted(init) == sorted(ac.state_dict())                 # incl. the `encoder.*` alias of the env-factor encoder
    ac.load_state_dict(init)
    with torch.no_grad():
        obs, priv, hist = (torch.from_numpy(d[k]) for k in ("probe_obs", "probe_priv", "probe_hist"))
        np.testing.assert_allclose(ac.act_teacher(obs, priv).numpy(), d["probe_teacher"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(ac.act_student(obs, hist).numpy(), d["probe_student"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(ac.evaluate(obs, priv).numpy(), d["probe_value"], rtol=1e-5, atol=1e-6)
        info = {}
        ac.act_inference({"obs": obs, "privileged_obs": priv, "obs_history": hist}, policy_info=info)
        assert info["gt_latents"].shape == (7, 4)      # (the student latent goes to act_student's own default dict, as upstream)
    alg = PPO(ac, device="cpu")
    alg.init_storage(N, T, [no], [npv], [no * H], [na])
    st = alg.storage
    for k in ("observations", "privileged_observations", "observation_histories", "actions", "rewards", "dones", "values", "mu", "sigma",
              "actions_log_prob"):
        getattr(st, k).copy_(torch.from_numpy(d["in_" + k]))
    st.step = T
    st.compute_returns(torch.from_numpy(d["last_values"]), PPO_Args.gamma, PPO_Args.lam)
    np.testing.assert_allclose(st.returns.numpy(), d["out_returns"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(st.advantages.numpy(), d["out_advantages"], rtol=1e-5, atol=1e-5)
    torch.manual_seed(int(d["seed"]) + 2)                          # same randperm as the reference run
    losses = alg.update()
    np.testing.assert_allclose(losses, d["losses"], rtol=2e-4, atol=1e-6)
    assert alg.learning_rate == pytest.approx(float(d["final_lr"]), rel=1e-6)
    for k, v in ac.state_dict().items():
        np.testing.assert_allclose(v.numpy(), d["final_" + k], rtol=2e-3, atol=2e-5, err_msg=k)


def test_teacher_student_rollout_surface():
    """act / process_env_step / compute_returns / update through the public surface, with the time-out bootstrap."""
    from go1_gym_learn.ppo import ActorCritic, RunnerArgs, caches, class_to_dict
    from go1_gym_learn.ppo.ppo import PPO, PPO_Args
    torch.manual_seed(0)
    ac = ActorCritic(70, 18, 2100, 12)
    alg = PPO(ac)
    alg.init_storage(8, 4, [70], [18], [2100], [12])
    for t in range(4):
        o, p, h = torch.randn(8, 70), torch.randn(8, 18), torch.randn(8, 2100)
        a = alg.act(o, p, h)
        assert a.shape == (8, 12)
        values = alg.transition.values.clone()
        rew = torch.randn(8)
        time_outs = torch.tensor([1, 0, 0, 0, 0, 0, 0, 1], dtype=torch.bool)
        alg.process_env_step(rew, time_outs.byte(), {"env_bins": torch.zeros(8), "time_outs": time_outs})
        expect = rew + PPO_Args.gamma * values[:, 0] * time_outs
        torch.testing.assert_close(alg.storage.rewards[t, :, 0], expect)
    alg.compute_returns(o, p)
    losses = alg.update()
    assert len(losses) == 3 and all(np.isfinite(losses)) and alg.storage.step == 0
    assert "sysid_residual" in caches.slot_cache.get_summary()
    assert class_to_dict(RunnerArgs)["num_steps_per_env"] == 24


@pytest.mark.parametrize("which", ["ppo_cse", "ppo"])
def test_storage_keeps_the_observations_the_policy_acted_on(which, small_ac_args, small_rma_args):
    """The HIP environment returns the SAME obs / privileged_obs tensors from every step and overwrites them in place
    (the reference allocates fresh ones, legged_robot.py:320-338).  Slot s of the storage must hold o_s — what act() saw —
    not o_{s+1}; then pi_old(a_s | stored o_s) reproduces the stored log-prob, i.e. the first-epoch PPO ratio is exactly 1."""
    torch.manual_seed(1)
    N, T, no, npv, H, na = 6, 5, 7, 3, 4, 12
    if which == "ppo_cse":
        from go1_gym_learn.ppo_cse.actor_critic import ActorCritic
        from go1_gym_learn.ppo_cse.ppo import PPO, gaussian_log_prob
    else:
        from go1_gym_learn.ppo import ActorCritic
        from go1_gym_learn.ppo.actor_critic import AC_Args
        from go1_gym_learn.ppo.ppo import PPO
        from go1_gym_learn.ppo_cse.ppo import gaussian_log_prob
        AC_Args.env_factor_encoder_branch_input_dims = [npv]
    ac = ActorCritic(no, npv, no * H, na)
    alg = PPO(ac, device="cpu")
    alg.init_storage(N, T, [no], [npv], [no * H], [na])
    obs_buf, priv_buf, hist_buf = torch.zeros(N, no), torch.zeros(N, npv), torch.zeros(N, no * H)      # the env's live buffers
    seen = []
    for s in range(T):
        obs_buf.normal_(); priv_buf.normal_(); hist_buf.normal_()          # "env.step" writes in place
        seen.append((obs_buf.clone(), priv_buf.clone(), hist_buf.clone()))
        alg.act(obs_buf, priv_buf, hist_buf)
        obs_buf.fill_(float("nan")); priv_buf.fill_(float("nan")); hist_buf.fill_(float("nan"))      # the step overwrites them
        alg.process_env_step(torch.randn(N), torch.zeros(N, dtype=torch.uint8), {"env_bins": torch.zeros(N)})
    st = alg.storage
    for s, (o, p, h) in enumerate(seen):
        assert torch.equal(st.observations[s], o) and torch.equal(st.privileged_observations[s], p), s
        assert torch.equal(st.observation_histories[s][:, :no * H].float(), h), s
    with torch.no_grad():
        for s, (o, p, h) in enumerate(seen):
            if which == "ppo_cse":
                mean, _, _ = alg.policy.forward(alg.body, st.observation_histories[s])
                logp = gaussian_log_prob(st.actions[s], mean.float(), alg.std)
            else:
                mean = ac.actor_body(torch.cat((st.observations[s], ac.env_factor_encoder(st.privileged_observations[s])), dim=-1))
                logp = gaussian_log_prob(st.actions[s], mean, ac.std)
            ratio = torch.exp(logp - st.actions_log_prob[s, :, 0])
            torch.testing.assert_close(ratio, torch.ones(N), rtol=1e-5, atol=1e-5)


def test_trajectory_helpers_match_reference():
    """go1_gym_learn.utils against the reference's functions (traj_utils.npz; recurrent-policy helpers, unused by ppo_cse)."""
    import os
    import numpy as np
    from util import GOLDEN
    from go1_gym_learn.utils import split_and_pad_trajectories, unpad_trajectories
    d = np.load(os.path.join(GOLDEN, "traj_utils.npz"))
    padded, masks = split_and_pad_trajectories(torch.from_numpy(d["x"]), torch.from_numpy(d["dones"]))
    assert np.array_equal(padded.numpy(), d["padded"]) and np.array_equal(masks.numpy(), d["masks"])
    assert np.array_equal(unpad_trajectories(padded, masks).numpy(), d["back"]) and np.array_equal(d["back"], d["x"])


def test_recurrent_mini_batches_match_reference():
    """`RolloutStorage.reccurent_mini_batch_generator` against the reference class (recurrent_batches.npz; an interface the
    feed-forward policy never uses)."""
    import os
    import numpy as np
    from util import GOLDEN
    from go1_gym_learn.ppo_cse.rollout_storage import RolloutStorage
    d = np.load(os.path.join(GOLDEN, "recurrent_batches.npz"))
    T, N = d["in_dones"].shape[:2]
    st = RolloutStorage(N, T, [7], [2], [21], [3], device="cpu")
    for name in ("observations", "privileged_observations", "observation_histories", "actions", "values", "advantages", "returns",
                 "actions_log_prob", "mu", "sigma", "dones"):
        getattr(st, name).copy_(torch.from_numpy(d["in_" + name]))
    batches = list(st.reccurent_mini_batch_generator(2, num_epochs=2))
    assert len(batches) == 4
    for i, batch in enumerate(batches):
        assert len(batch) == 12
        for j, t in enumerate(batch):
            assert np.array_equal(t.numpy(), d[f"b{i}_{j}"]), (i, j)


def test_rollout_path_matches_reference(small_ac_args):
    """PPO.act -> process_env_step x T -> compute_returns against the reference classes on the same observation streams and the
    same global-generator seed (rollout.npz): sampled actions, log-probabilities, values, the time-out bootstrap of the rewards
    (ppo.py:83-86), what the storage holds, returns and advantages."""
    from go1_gym_learn.ppo_cse.actor_critic import ActorCritic
    from go1_gym_learn.ppo_cse.ppo import PPO
    d = np.load(os.path.join(GOLDEN, "rollout.npz"))
    N, T, no, npv, H, na = [int(x) for x in d["dims"]]
    ac = ActorCritic(no, npv, no * H, na)
    ac.load_state_dict({k[5:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("init_")})
    alg = PPO(ac, device="cpu")
    alg.init_storage(N, T, [no], [npv], [no * H], [na])
    t_ = lambda k: torch.from_numpy(d[k])
    torch.manual_seed(int(d["seed"]) + 2)
    for t in range(T):
        a = alg.act(t_("obs")[t], t_("priv")[t], t_("hist")[t])
        np.testing.assert_allclose(a.numpy(), d["acts"][t], rtol=1e-5, atol=1e-6)
        alg.process_env_step(t_("rew")[t].clone(), t_("dones_in")[t].clone(), {"env_bins": t_("bins")[t], "time_outs": t_("touts")[t]})
    alg.compute_returns(t_("hist")[T], t_("priv")[T])
    st = alg.storage
    for k in ("observations", "privileged_observations", "actions", "rewards", "values", "actions_log_prob", "mu", "sigma", "env_bins",
              "returns", "advantages"):
        np.testing.assert_allclose(getattr(st, k).numpy(), d["st_" + k], rtol=2e-5, atol=2e-5, err_msg=k)
    assert np.array_equal(st.dones.numpy().astype(bool), d["st_dones"].astype(bool))
    np.testing.assert_allclose(st.observation_histories[..., :no * H].float().numpy(), d["st_observation_histories"], rtol=1e-6, atol=1e-6)
    assert np.abs(d["st_rewards"][..., 0] - d["rew"]).max() > 0.05                       # the bootstrap did act
