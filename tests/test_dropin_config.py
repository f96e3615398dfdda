"""The reference's scripts/train.py must run unchanged on top of this package's modules.

Executed here (authoring container, where /root/reference exists): the reference script's `train_go1` is run
verbatim with our `go1_gym`, `go1_gym_learn`, `params_proto`/`ml_logger`/`isaacgym` stand-ins on sys.path; the
env constructor is intercepted (no GPU here) and the `Cfg` it would have been built from is compared with the
table bench.py / smoke() use (walk-these-ways_amd/scripts/train_config.py).  Skipped where the reference tree is
absent (GPU box)."""
import importlib.util
import os
import sys

import pytest

REF_TRAIN = "/root/reference/scripts/train.py"


@pytest.mark.skipif(not os.path.exists(REF_TRAIN), reason="reference tree not present")
def test_reference_train_script_configures_our_env(monkeypatch):
    import go1_gym.envs.go1.velocity_tracking as vt
    import go1_gym_learn.ppo_cse as runner_mod
    from go1_gym.envs.base import legged_robot_config
    from go1_gym.envs.base.legged_robot_config import make_cfg
    from scripts.train_config import apply_train_config
    import go1sim_host as H
    from ml_logger import logger

    fresh = make_cfg()
    monkeypatch.setattr(legged_robot_config, "Cfg", fresh)
    captured = {}

    class FakeEnv:
        def __init__(self, sim_device, headless, cfg=None, **kw):
            captured.update(sim_device=sim_device, cfg=cfg)
            raise StopIteration           # stop train_go1 right after the env would have been constructed

    monkeypatch.setattr(vt, "VelocityTrackingEasyEnv", FakeEnv)
    spec = importlib.util.spec_from_file_location("ref_train", REF_TRAIN)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)          # imports only; __main__ block does not run
    logger.configure("test", root="/tmp/wtw_logger_test")
    with pytest.raises(StopIteration):
        mod.train_go1(headless=True)
    assert captured["sim_device"] == "cuda:0"
    ref_cfg = captured["cfg"]
    ours = apply_train_config(make_cfg())
    assert vars(ref_cfg) == vars(ours)
    # and the flattened simulator configuration is byte-identical
    S1, m1 = H.build_sim_config(ref_cfg, num_envs=64)
    S2, m2 = H.build_sim_config(ours, num_envs=64)
    assert bytes(S1) == bytes(S2)
    assert m1["reward_names"] == m2["reward_names"] and len(m1["reward_names"]) == 19
    assert (S1.max_episode_length, S1.resample_interval, S1.rand_interval, S1.gravity_rand_interval,
            S1.gravity_rand_duration) == (1001, 500, 201, 401, 397)
    assert hasattr(runner_mod, "Runner") and hasattr(runner_mod, "RunnerArgs")


def test_cpu_device_is_refused_loudly():
    from go1_gym.envs.base.legged_robot_config import make_cfg
    from go1_gym.envs.go1.velocity_tracking import VelocityTrackingEasyEnv
    from scripts.train_config import apply_train_config
    cfg = apply_train_config(make_cfg(), num_envs=4)
    with pytest.raises(RuntimeError, match="no CPU simulation path"):
        VelocityTrackingEasyEnv(sim_device="cpu", headless=True, cfg=cfg)


@pytest.mark.skipif(not os.path.exists(REF_TRAIN), reason="reference tree not present")
def test_reference_train_script_runs_one_iteration_end_to_end(monkeypatch, tmp_path):
    """The reference's `train_go1` body, verbatim, THROUGH the constructor and into `Runner(env).learn(...)`
    (train.py:207-216): env construction, HistoryWrapper, logger.log_params / log_text, Runner with RunnerArgs,
    one full PPO iteration (rollout, GAE, update, metrics, checkpoint + TorchScript export).  No GPU in this container:
    the simulator handle is the oracle-backed stand-in (tests/fake_sim.py) on CPU buffers, the env is shrunk to 48
    robots and the Runner is pointed at 'cpu'; every class in between is the product's.  Then the reference's scripts/play.py
    on the run directory this produced: `load_policy` on the exported TorchScript files and `play_go1` verbatim (configuration
    read back from the logged parameters.pkl, one environment, 250 policy steps)."""
    import fake_sim
    import go1_gym.envs.go1.velocity_tracking as vt
    import go1_gym_learn.ppo_cse as runner_mod
    from go1_gym.envs.base import legged_robot_config
    from go1_gym.envs.base.legged_robot_config import make_cfg
    from ml_logger import logger

    monkeypatch.setattr(legged_robot_config, "Cfg", make_cfg())
    fake_sim.install(monkeypatch)
    real_env, real_runner = vt.VelocityTrackingEasyEnv, runner_mod.Runner
    seen = {}

    def small_env(sim_device, headless, cfg=None, **kw):
        seen["sim_device"] = sim_device
        return real_env(sim_device, headless, num_envs=48, cfg=cfg, **kw)

    class CpuRunner(real_runner):
        def __init__(self, env, device="cpu"):
            seen["runner_device"] = device
            super().__init__(env, device="cpu")

        def learn(self, num_learning_iterations, **kw):
            seen["learn_kwargs"] = dict(num_learning_iterations=num_learning_iterations, **kw)
            return super().learn(1, **kw)

    monkeypatch.setattr(vt, "VelocityTrackingEasyEnv", small_env)
    monkeypatch.setattr(runner_mod, "Runner", CpuRunner)
    spec = importlib.util.spec_from_file_location("ref_train_full", REF_TRAIN)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    logger.configure("dropin", root=str(tmp_path))
    logger.print_summary = False
    monkeypatch.chdir(tmp_path)
    old = (runner_mod.RunnerArgs.save_video_interval,)
    try:
        mod.train_go1(headless=True)
    finally:
        (runner_mod.RunnerArgs.save_video_interval,) = old
    assert seen["sim_device"] == "cuda:0" and seen["runner_device"] == "cuda:0"
    assert seen["learn_kwargs"] == dict(num_learning_iterations=100000, init_at_random_ep_len=True, eval_freq=100)   # train.py:216
    ck = tmp_path / "dropin" / "checkpoints"
    assert (ck / "ac_weights_last.pt").exists() and (ck / "adaptation_module_latest.jit").exists() and (ck / "body_latest.jit").exists()
    from util import check_exported_policy_layout
    check_exported_policy_layout(str(ck))
    # ... and the reference's own play.py loader (`load_policy`, play.py:17-29) turns that run directory into a policy
    import torch
    spec = importlib.util.spec_from_file_location("ref_play", "/root/reference/scripts/play.py")
    play = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(play)
    policy = play.load_policy(str(tmp_path / "dropin"))
    info = {}
    act = policy({"obs_history": torch.randn(5, 2100)}, info)
    assert act.shape == (5, 12) and info["latent"].shape == (5, 2) and bool(torch.isfinite(act).all())
    # ... the REFERENCE ActorCritic loads the weights file strictly (its own interpreter: the package names collide)
    import subprocess
    code = ("import sys, torch; sys.path[:0] = [%r, '/root/reference']; "
            "from go1_gym_learn.ppo_cse.actor_critic import ActorCritic; import go1_gym_learn; "
            "assert go1_gym_learn.__file__.startswith('/root/reference'); ac = ActorCritic(70, 2, 2100, 12); "
            "r = ac.load_state_dict(torch.load(%r, map_location='cpu'), strict=True); print('LOADED', r)"
            % (os.path.join(os.path.dirname(__file__), "..", "walk-these-ways_amd", "shims"), str(ck / "ac_weights_last.pt")))
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "LOADED <All keys matched successfully>" in res.stdout, res.stdout[-500:] + res.stderr[-1500:]
    # ... and `play_go1` verbatim (play.py:89-157): the run directory found by its glob, `parameters.pkl` as train.py logged it
    # read back into Cfg, one environment, the policy in the loop for 250 steps, the two plots (Agg backend)
    import matplotlib
    matplotlib.use("Agg")
    run = tmp_path / "runs" / "gait-conditioned-agility" / "pretrain-v0" / "train"
    run.mkdir(parents=True)
    os.symlink(tmp_path / "dropin", run / "000000.000000")
    (tmp_path / "scripts").mkdir()
    monkeypatch.chdir(tmp_path / "scripts")
    monkeypatch.setattr(vt, "VelocityTrackingEasyEnv", real_env)           # (play.py builds its single environment itself)
    monkeypatch.setattr(play, "VelocityTrackingEasyEnv", real_env)
    monkeypatch.setattr(play, "tqdm", lambda it: it)
    fresh = make_cfg()                                                     # (play.py runs in its own process: a pristine Cfg)
    monkeypatch.setattr(legged_robot_config, "Cfg", fresh)
    monkeypatch.setattr(play, "Cfg", fresh)
    play.play_go1(headless=True)
    monkeypatch.chdir(tmp_path)
    metrics = logger.load_pkl("metrics.pkl")
    assert metrics and metrics[-1]["timesteps"] == 24 * 48
    assert any(k.startswith("train/episode/rew_") for k in metrics[-1])


REF_TEST = "/root/reference/scripts/test.py"


@pytest.mark.skipif(not os.path.exists(REF_TEST), reason="reference tree not present")
def test_reference_test_script_runs_verbatim(monkeypatch):
    """BASELINE configs[0] semantics: the reference's scripts/test.py `run_env` verbatim — config, env construction,
    `env.reset()`, 1000 steps of zero actions with the 4-tuple return (test.py:188-200) — on the oracle-backed stand-in."""
    import fake_sim
    import torch
    import go1_gym.envs.go1.velocity_tracking as vt
    from go1_gym.envs.base import legged_robot_config
    from go1_gym.envs.base.legged_robot_config import make_cfg

    monkeypatch.setattr(legged_robot_config, "Cfg", make_cfg())
    fake_sim.install(monkeypatch)
    made = []
    real_env = vt.VelocityTrackingEasyEnv

    def env_factory(*a, **kw):
        made.append(real_env(*a, **kw))
        return made[-1]
    monkeypatch.setattr(vt, "VelocityTrackingEasyEnv", env_factory)
    spec = importlib.util.spec_from_file_location("ref_test_script", REF_TEST)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.run_env(render=False, headless=True)
    env = made[0]
    assert env.num_envs == 3 and env.common_step_counter == 1001          # reset()'s step + 1000
    assert env.obs_buf.shape == (3, env.num_obs) and env.obs_buf.dtype == torch.float32
    for t in (env.obs_buf, env.rew_buf, env.root_states, env.dof_pos, env.contact_forces):
        assert torch.isfinite(t).all()
    assert float(env.root_states[:, 2].min()) > 0.15                      # zero actions: the robots stand


def test_every_reference_module_path_on_the_hot_path_resolves():
    """the reference's package tree (go1_gym, go1_gym_learn) module by module — same dotted paths, same public names.  Not
    mirrored: go1_gym_learn.eval_metrics (offline evaluation sweeps, outside SURVEY §8)."""
    import importlib
    surface = {
        "go1_gym": ["MINI_GYM_ROOT_DIR"],
        "go1_gym.envs.base.base_task": ["BaseTask"],
        "go1_gym.envs.base.curriculum": ["Curriculum", "RewardThresholdCurriculum"],
        "go1_gym.envs.base.legged_robot": ["LeggedRobot"],
        "go1_gym.envs.base.legged_robot_config": ["Cfg"],
        "go1_gym.envs.go1.go1_config": ["config_go1"],
        "go1_gym.envs.go1.velocity_tracking": ["VelocityTrackingEasyEnv"],
        "go1_gym.envs.rewards.corl_rewards": ["CoRLRewards"],
        "go1_gym.envs.wrappers.history_wrapper": ["HistoryWrapper"],
        "go1_gym.utils.math_utils": ["quat_apply_yaw", "wrap_to_pi", "get_scale_shift"],
        "go1_gym.utils.terrain": ["Terrain"],
        "go1_gym_learn.env": ["VecEnv"],
        "go1_gym_learn.env.vec_env": ["VecEnv"],
        "go1_gym_learn.utils": ["split_and_pad_trajectories", "unpad_trajectories"],
        "go1_gym_learn.ppo": ["Runner", "RunnerArgs"],
        "go1_gym_learn.ppo.actor_critic": ["ActorCritic", "AC_Args"],
        "go1_gym_learn.ppo.metrics_caches": ["DistCache", "SlotCache"],
        "go1_gym_learn.ppo.ppo": ["PPO", "PPO_Args"],
        "go1_gym_learn.ppo.rollout_storage": ["RolloutStorage"],
        "go1_gym_learn.ppo_cse": ["Runner", "RunnerArgs"],
        "go1_gym_learn.ppo_cse.actor_critic": ["ActorCritic", "AC_Args"],
        "go1_gym_learn.ppo_cse.metrics_caches": ["DistCache", "SlotCache"],
        "go1_gym_learn.ppo_cse.ppo": ["PPO", "PPO_Args"],
        "go1_gym_learn.ppo_cse.rollout_storage": ["RolloutStorage"],
    }
    for mod, names in surface.items():
        m = importlib.import_module(mod)
        for n in names:
            assert hasattr(m, n), (mod, n)


def _tree(node):
    import inspect
    out = {}
    for k in dir(node):
        if k.startswith("_"):
            continue
        v = getattr(node, k)
        if inspect.isclass(v):
            out[k] = _tree(v)
        elif callable(v):
            continue
        else:
            out[k] = list(v) if isinstance(v, tuple) else v
    return out


def _tree_diff(mine, ref, path=""):
    out = []
    for k in sorted(set(mine) | set(ref)):
        if k not in mine:
            out.append((path + k, "<absent>", ref[k]))
        elif k not in ref:
            out.append((path + k, mine[k], "<absent>"))
        elif isinstance(mine[k], dict) and isinstance(ref[k], dict):
            out += _tree_diff(mine[k], ref[k], path + k + ".")
        elif mine[k] != ref[k]:
            out.append((path + k, mine[k], ref[k]))
    return out


def test_every_configuration_default_equals_the_reference():
    """`Cfg` as declared, `Cfg` after `config_go1`, and AC_Args / PPO_Args / RunnerArgs of both learners, leaf by leaf against
    the reference's classes (cfg_defaults.json).  Allowed: keys this repo ADDS (MI355X options, all inert by default)."""
    import importlib
    import json
    import os
    from util import GOLDEN
    from go1_gym.envs.base.legged_robot_config import make_cfg
    from go1_gym.envs.go1.go1_config import config_go1
    with open(os.path.join(GOLDEN, "cfg_defaults.json")) as f:
        ref = json.load(f)
    cfg = make_cfg()
    diffs = {"Cfg": _tree_diff(_tree(cfg), ref["Cfg"])}
    config_go1(cfg)
    diffs["Cfg_go1"] = _tree_diff(_tree(cfg), ref["Cfg_go1"])
    for pkg in ("ppo", "ppo_cse"):
        ac = importlib.import_module(f"go1_gym_learn.{pkg}.actor_critic")
        pp = importlib.import_module(f"go1_gym_learn.{pkg}.ppo")
        rn = importlib.import_module(f"go1_gym_learn.{pkg}")
        for name, cls in (("AC_Args", ac.AC_Args), ("PPO_Args", pp.PPO_Args), ("RunnerArgs", rn.RunnerArgs)):
            diffs[f"{pkg}.{name}"] = _tree_diff(_tree(cls), ref[pkg][name])
    wrong = {k: [d for d in v if d[2] != "<absent>"] for k, v in diffs.items()}          # a value that differs or a key we lack
    assert not any(wrong.values()), wrong
    added = sorted({d[0] for v in diffs.values() for d in v if d[2] == "<absent>"})
    print("keys added by this repo:", added)


@pytest.mark.parametrize("fname", ["runner_iteration.npz", "runner_iteration_eval.npz"])
def test_runner_reproduces_the_reference_runner_end_to_end(monkeypatch, tmp_path, fname):
    """Two learning iterations of the product's `Runner` (restructured PPO: fused first layer, flat gradient buffer, explicit
    Gaussian algebra, lazy episode statistics) against the REFERENCE Runner / PPO / ActorCritic / RolloutStorage driving the same
    environment from the same seeds (tests/golden/runner_iteration.npz, gen_runner_iteration.py): same last rollout, same
    learning rate, same weights.  Episodes end and restart inside the run (time-outs with value bootstrap, falls)."""
    import json
    import sys
    import numpy as np
    import torch
    import fake_sim
    from util import GOLDEN
    sys.path.insert(0, GOLDEN)
    import gen_runner_iteration as G
    from go1_gym_learn.ppo_cse import Runner, RunnerArgs
    from go1_gym_learn.ppo_cse.actor_critic import AC_Args
    from ml_logger import logger
    d = np.load(os.path.join(GOLDEN, fname))                      # (the second: 16 evaluation environments behind the 32 training ones)
    st_ = json.loads(str(d["settings"]))
    fake_sim.install(monkeypatch)
    env = G.build_env(st_)
    logger.configure("runner_iteration", root=str(tmp_path))
    logger.print_summary = False
    monkeypatch.chdir(tmp_path)
    for k, v in (("actor_hidden_dims", st_["actor"]), ("critic_hidden_dims", st_["critic"]), ("adaptation_module_branch_hidden_dims", st_["adaptation"])):
        monkeypatch.setattr(AC_Args, k, v)
    monkeypatch.setattr(RunnerArgs, "num_steps_per_env", st_["num_steps_per_env"])
    monkeypatch.setattr(RunnerArgs, "save_video_interval", 0)
    torch.manual_seed(st_["seed"] + 1)
    runner = Runner(env, device="cpu")
    for k, v in runner.alg.sync_module().state_dict().items():
        assert np.array_equal(v.numpy(), d["init_" + k]), k                     # same initialisation from the same seed
    runner.learn(num_learning_iterations=st_["iterations"], init_at_random_ep_len=False, eval_freq=100)
    st = runner.alg.storage
    assert int(d["last_dones"].sum()) > 0 and runner.tot_timesteps == int(d["tot_timesteps"])
    assert np.array_equal(st.dones.numpy().astype(bool), d["last_dones"].astype(bool))
    for k, tol in (("actions", 1e-4), ("rewards", 1e-4), ("values", 1e-4), ("returns", 1e-4), ("advantages", 1e-3)):
        np.testing.assert_allclose(getattr(st, k).numpy(), d["last_" + k], rtol=tol, atol=tol, err_msg=k)
    assert runner.alg.learning_rate == pytest.approx(float(d["lr"]), rel=1e-6)
    for k, v in runner.alg.sync_module().state_dict().items():
        np.testing.assert_allclose(v.numpy(), d["final_" + k], rtol=1e-3, atol=1e-4, err_msg=k)


REF_WRAPPER = "/root/reference/go1_gym/envs/wrappers/history_wrapper.py"


@pytest.mark.skipif(not os.path.exists(REF_WRAPPER), reason="reference tree not present")
def test_reference_history_wrapper_over_this_environment(monkeypatch):
    """The REFERENCE `HistoryWrapper` class (history_wrapper.py:6-41, `torch.cat` of the history every call) wrapped around this
    repository's environment, next to the product's wrapper (a window of the ring the simulator appends to) around an identical
    second environment: same observation dicts call by call — reset, the extra shift of get_observations, steps across episode
    ends (the history is not cleared when an environment resets)."""
    import numpy as np
    import torch
    import fake_sim
    from go1_gym.envs.base.legged_robot_config import make_cfg
    from go1_gym.envs.go1.velocity_tracking import VelocityTrackingEasyEnv
    from go1_gym.envs.wrappers.history_wrapper import HistoryWrapper
    from scripts.train_config import apply_train_config
    fake_sim.install(monkeypatch)
    spec = importlib.util.spec_from_file_location("_ref_history_wrapper", REF_WRAPPER)
    ref_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_mod)

    def build():
        cfg = apply_train_config(make_cfg(), num_envs=16)
        cfg.terrain.mesh_type = "plane"
        cfg.env.episode_length_s = 0.2                 # episodes end inside the run
        cfg.env.num_observation_history = 5
        torch.manual_seed(0)
        return VelocityTrackingEasyEnv(sim_device="cuda:0", headless=True, cfg=cfg)
    mine, ref = HistoryWrapper(build()), ref_mod.HistoryWrapper(build())
    assert (mine.num_obs_history, mine.obs_history_length) == (ref.num_obs_history, ref.obs_history_length) == (350, 5)
    g = torch.Generator().manual_seed(1)
    resets = 0
    for call in ["reset", "get_observations"] + ["step"] * 7 + ["get_observations"] + ["step"] * 9 + ["reset", "step", "step"]:
        if call == "step":
            a = 0.5 * torch.randn(16, 12, generator=g)
            (om, rm, dm, _), (orf, rr, dr, _) = mine.step(a.clone()), ref.step(a.clone())
            assert torch.equal(rm, rr) and torch.equal(dm, dr)
            resets += int(dm.sum())
        else:
            om, orf = getattr(mine, call)(), getattr(ref, call)()
        for k in ("obs", "privileged_obs", "obs_history"):
            assert torch.equal(om[k], orf[k]), (call, k, float((om[k] - orf[k]).abs().max()))
    assert resets > 0
