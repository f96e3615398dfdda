"""PPO with adaptive-KL learning rate, clipped value loss and an adaptation-module regression step
(mirror of reference go1_gym_learn/ppo_cse/ppo.py:13-205), PyTorch-ROCm.

Same algorithm, hyper-parameters (`PPO_Args`) and public methods (`act`, `process_env_step`,
`compute_returns`, `update`).  What is organised differently for the MI355X:

  * mixed precision the classic way instead of per-op autocast: fp32 master weights (the exported / checkpointed
    `actor_critic`) + a bf16 compute replica; one multi-tensor copy per optimiser step in each direction, no cast
    kernels around every Linear;
  * one flat fp32 gradient buffer: master `.grad`s are views into it, so global-norm clipping is one norm over one
    tensor and the data-parallel all-reduce (RCCL over xGMI, only when envs are partitioned per GPU) is one
    collective on that buffer per optimiser step — no bucketing, no DDP hooks;
  * the Gaussian policy algebra (log-prob, entropy, KL, surrogate, clipped value loss) is written out on (M, 12)
    tensors instead of going through torch.distributions (which validates `std >= 0` with a host sync per sample);
  * no host synchronisation inside `update()`: the KL-adaptive learning rate is a device tensor handed to Adam,
    loss statistics are accumulated on device and read once at the end.
"""
import math
import os
import sys

import torch
import torch.distributed as dist
import torch.nn.functional as F
import torch.optim as optim
from params_proto import PrefixProto

from go1_gym_learn.ppo_cse import ActorCritic
from go1_gym_learn.ppo_cse import RolloutStorage
from go1_gym_learn.ppo_cse import caches  # noqa: F401
from go1_gym_learn.ppo_cse.flat_policy import FlatPolicy


class PPO_Args(PrefixProto):
    value_loss_coef = 1.0
    use_clipped_value_loss = True
    clip_param = 0.2
    entropy_coef = 0.01
    num_learning_epochs = 5
    num_mini_batches = 4
    learning_rate = 1.e-3
    adaptation_module_learning_rate = 1.e-3
    num_adaptation_module_substeps = 1
    schedule = 'adaptive'
    gamma = 0.99
    lam = 0.95
    desired_kl = 0.01
    max_grad_norm = 1.
    selective_adaptation_module_loss = False
    # MI355X additions
    autocast_bf16 = False           # BASELINE config 2: "bf16 policy" (fp32 master weights + bf16 compute replica)
    data_parallel = True            # all-reduce gradients when torch.distributed is initialised with world_size > 1
    dp_grad_dtype = "fp32"          # "bf16": the PPO-stage gradient crosses the links in bf16 (half the bytes of the 13 MB exchange)
    dp_zero1 = False                # PPO stage: reduce-scatter the gradient, every rank steps its 1/world slice of the flat
                                    # parameter (global norm / KL by two scalar-sized all-reduces), all-gather the result
    history_ring = True             # keep each observation once when init_storage(..., sliding_history=True) allows it
    # replay the mini-batch step as a HIP graph from the second update() on.  True: the fused update only (its kernels are this
    # repo's own).  "all": the autograd update too — NOT safe at production batch sizes on ROCm 7.2: torch's multi-block
    # reductions (bias / std gradients, column sums over 24576 rows) zero their semaphores with hipMemsetAsync, and that memset
    # node does not replay reliably (wrong gradients from the first pure replay on; DESIGN.md "HIP graphs", tools/probes/
    # graph_reduce_repro.py) — exact at the test sizes (<= 1024 rows, single-block reductions), hence still available for them.
    use_hip_graphs = True
    dp_graph_collectives = True     # data parallel: record the RCCL collectives of a mini-batch INSIDE its HIP graph (one replay per
                                    # mini-batch, no eager launches between graphs); if the capture fails the three-graph scheme with
                                    # eager collectives between the graphs is used instead
    use_fused_kernels = True        # bf16 policy on a GPU: hand-scheduled forward/backward with csrc/go1ppo.hip (fused.py)
    use_tuned_gemms = True          # PyTorch TunableOp with the gfx950 table shipped in walk-these-ways_amd/tuning/


_HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)


def _enable_tuned_gemms():
    """hipBLASLt's default heuristics pick poor kernels for this policy's tall-skinny bf16 shapes (M = 24576 rows,
    N/K = 64..2104); TunableOp selections for gfx950 ship in walk-these-ways_amd/tuning/ (shapes that are not in the
    table are tuned once on first use).  Environment variables (PYTORCH_TUNABLEOP_*) take precedence."""
    import os
    if "PYTORCH_TUNABLEOP_ENABLED" in os.environ:
        return
    try:
        from torch.cuda import tunable
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tuning", "tunableop_gfx950.csv")
        tunable.enable(True)
        tunable.set_filename(os.path.normpath(path), insert_device_ordinal=False)      # one table for every rank
        tunable.set_max_tuning_duration(30)
        if hasattr(tunable, "write_file_on_exit"):
            tunable.write_file_on_exit(False)       # the shipped table is read-only; tools/tune_gemms.sh regenerates it
    except Exception as err:          # an optimisation only
        print(f"[ppo] TunableOp not enabled ({type(err).__name__}: {err})", file=sys.stderr)


_AUTOCAST_DEFAULT = bool(PPO_Args.autocast_bf16)          # the class default, before any caller touched it


def policy_dtype_is_bf16():
    """bf16 policy (fp32 master + bf16 compute copy, the fused update) or the fp32 autograd path?  `PPO_Args.autocast_bf16` decides,
    unless the environment variable GO1_POLICY_DTYPE (= bf16 | fp32) is set: the switch for the UNCHANGED reference scripts
    (scripts/train.py constructs Runner with the default PPO_Args, train.py:207-216) — `GO1_POLICY_DTYPE=bf16 python scripts/train.py`
    selects BASELINE configs[1]'s bf16 policy without editing the script."""
    v = os.environ.get("GO1_POLICY_DTYPE", "").strip().lower()
    if v and v not in ("bf16", "bfloat16", "fp32", "f32", "float32"):
        raise ValueError(f"GO1_POLICY_DTYPE={v!r}: expected bf16 or fp32")
    if v:
        want = v in ("bf16", "bfloat16")
        # the variable speaks for scripts that leave PPO_Args at its class default; a caller that SET autocast_bf16 (bench.py --fp32, a test)
        # and meets a contradicting variable gets an error instead of a silently different dtype
        explicit = bool(PPO_Args.autocast_bf16) != _AUTOCAST_DEFAULT
        if explicit and bool(PPO_Args.autocast_bf16) != want:
            raise ValueError(f"GO1_POLICY_DTYPE={v} contradicts PPO_Args.autocast_bf16={PPO_Args.autocast_bf16}")
        return want
    return bool(PPO_Args.autocast_bf16)


def _force_dp():
    return os.environ.get("GO1_FORCE_DP", "0") not in ("", "0")


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def gaussian_log_prob(actions, mean, std):
    """sum_j log N(a_j; mu_j, sigma_j) — what Normal(mean, std).log_prob(a).sum(-1) evaluates."""
    z = (actions - mean) / std
    return -0.5 * (z * z).sum(dim=-1) - (torch.log(std).sum() + actions.shape[-1] * _HALF_LOG_2PI)


def gaussian_entropy(std):
    """sum_j H[N(., sigma_j)]; identical for every sample because sigma does not depend on the observation."""
    return (0.5 + _HALF_LOG_2PI) * std.numel() + torch.log(std).sum()


class PPO:
    actor_critic: ActorCritic

    def __init__(self, actor_critic, device='cpu'):
        self.device = device
        self.actor_critic = actor_critic          # public / checkpoint format; refreshed by sync_module()
        self.actor_critic.to(device)
        self.on_gpu = torch.device(device).type == "cuda"
        self.bf16 = bool(policy_dtype_is_bf16() and self.on_gpu)
        self.storage = None
        ac = self.actor_critic
        self.policy = FlatPolicy(ac)
        n = self.policy.numel
        self.n_body, self.n_std = n, ac.std.numel()
        # fp32 master: [flat policy | std | pad]; the single tensor both optimisers step (length: a multiple of the world
        # size, so that it splits into equal slices for the sharded step)
        w = _world() if PPO_Args.data_parallel else 1
        self.master = torch.zeros(-(-(n + 16) // w) * w, device=device)
        self.policy.pack(ac, self.master[:n])
        self.master[n:n + self.n_std].copy_(ac.std.detach())
        self.master.grad = torch.zeros_like(self.master)
        # the mean KL of a mini-batch rides in the first padding slot behind the std gradient: the data-parallel
        # all-reduce of the flat gradient carries it along (one collective instead of two per optimiser step)
        self._kl = self.master.grad[n + self.n_std:n + self.n_std + 1].view(())
        self._g_live = self.master.grad[:n + self.n_std]          # what the norm / clip / optimiser see
        pol = self.policy
        self._priv_cols_grad = pol._block(self.master.grad[:n], "W1")[:pol.first[0] + pol.first[1], pol.K + 1:pol.K + 1 + pol.npv]
        # compute copies (leaves of the autograd graph): policy body in bf16 or fp32, std always fp32
        self.body = torch.zeros(n, device=device, dtype=torch.bfloat16 if self.bf16 else torch.float32).requires_grad_()
        self.std = torch.zeros(self.n_std, device=device).requires_grad_()
        kw = dict(fused=True, capturable=True) if self.on_gpu else {}
        self._acc = torch.zeros(4, device=device)        # value, surrogate, adaptation, adaptation-test losses
        self._idx_all, self._graphs, self._updates_done = None, {}, 0
        self._collectives_capturable = None      # unknown until the first data-parallel capture (True / False afterwards)
        self._pregathered, self._Xall = False, None
        lr = torch.tensor(PPO_Args.learning_rate, device=device) if self.on_gpu else PPO_Args.learning_rate
        self.optimizer = optim.Adam([self.master], lr=lr, **kw)
        # the reference builds this optimiser over all parameters (ppo.py:45-46) but only the adaptation module ever
        # receives a non-zero gradient from the adaptation loss; with exactly-zero gradients Adam's moments stay zero
        # and the other parameters do not move, so stepping the whole flat buffer is equivalent
        self.adaptation_module_optimizer = optim.Adam([self.master], lr=PPO_Args.adaptation_module_learning_rate, **kw)
        self.transition = RolloutStorage.Transition()
        self._lr = torch.tensor(PPO_Args.learning_rate, device=device)
        # GO1_FORCE_DP=1: take the data-parallel path in a process group of ONE rank too (every collective, the graph split and
        # the sharded step then execute — how the RCCL path is exercised on a single-GPU box, tests/test_gpu_distributed.py)
        self.dp = bool(PPO_Args.data_parallel and (_world() > 1 or (_force_dp() and dist.is_available() and dist.is_initialized())))
        # hand-scheduled forward/backward (fused.py): bf16 policy with ELU activations on a GPU.  No silent fallback:
        # when it applies and libgo1ppo.so is missing, load_library() raises.
        self.fused = bool(self.bf16 and PPO_Args.use_fused_kernels and self.policy.act is torch.nn.ELU)
        self._roll_net = self._train_net = None
        self._roll_noise = None           # (T, N, actions) sampling noise of the current rollout (fused path)
        self._update_graph_ok = True      # cleared when the whole-update graph could not be captured (update())
        self._opt = self._opt_ad = None
        if self.on_gpu and PPO_Args.use_tuned_gemms:
            _enable_tuned_gemms()
        self._ad_grad_views = [self.master.grad]      # what the adaptation stage's gradient all-reduce has to cover
        if self.fused:
            from go1_gym_learn.ppo_cse import fused
            self._fused_lib = fused.load_library()
            # optimiser steps as two kernels each (fused.FusedAdam); the adaptation optimiser only visits the
            # adaptation module's elements (all other gradients of that stage are exactly zero, see above)
            pol, ns = self.policy, self.n_std
            self._opt = fused.FusedAdam(self._fused_lib, self.master, self.body, self.std, n, PPO_Args.learning_rate, ranges=[(0, n + ns)])
            self._opt_ad = fused.FusedAdam(self._fused_lib, self.master, self.body, self.std, n, PPO_Args.adaptation_module_learning_rate,
                                           ranges=[(0, pol.adaptation_numel)])
            self._lr = self._opt.lr
            self._ad_grad_views = [self.master.grad[:pol.adaptation_numel]]
        if self.dp:                              # identical initial weights on every rank
            dist.broadcast(self.master, src=0)
        self._dp_lowp = self._dp_shard = None
        if self.dp:
            A = PPO_Args
            assert A.dp_grad_dtype in ("fp32", "bf16")
            if A.dp_grad_dtype == "bf16":
                self._dp_lowp = torch.zeros(self.master.numel(), device=device, dtype=torch.bfloat16)
            if A.dp_zero1:
                per = self.master.numel() // _world()
                self._dp_shard = (dist.get_rank() * per, per)
                self._dp_shard_buf = torch.zeros(per, device=device, dtype=torch.bfloat16 if self._dp_lowp is not None else torch.float32)
                self._dp_tail = torch.zeros(self.master.numel() - n, device=device)
                if self._opt is not None:
                    # the slice stops at the last live parameter: the KL slot and the padding behind it are carried by the
                    # flat buffer for the exchange only — stepping them would write past `std` and treat KL as a gradient
                    lo = self._dp_shard[0]
                    self._opt.set_ranges([(lo, max(0, min(lo + per, n + self.n_std) - lo))])
        self._push_weights()

    # ---- precision plumbing --------------------------------------------------------------------------
    @property
    def flat_param(self):
        return self.master

    @property
    def flat_grad(self):
        return self.master.grad

    def _push_weights(self):
        """fp32 master -> compute copies: one cast kernel + one tiny copy."""
        with torch.no_grad():
            self.body.copy_(self.master[:self.n_body])
            self.std.copy_(self.master[self.n_body:self.n_body + self.n_std])
            if self._train_net is not None:
                self._train_net.refresh_transposes()

    def _pull_grads(self):
        """compute-copy gradients -> fp32 master gradient."""
        g = self.master.grad
        if self.body.grad is not None:
            g[:self.n_body].copy_(self.body.grad)
        else:
            g[:self.n_body].zero_()
        if self.std.grad is not None:
            g[self.n_body:self.n_body + self.n_std].copy_(self.std.grad)
        else:
            g[self.n_body:].zero_()
        self.body.grad = None
        self.std.grad = None
        # the augmented rows carry the privileged observations for the critic only: the adaptation module and the
        # actor must not see them, so their (structurally zero) weights on those columns receive no gradient
        self._priv_cols_grad.zero_()

    def sync_module(self):
        """Refresh the nn.Module (state_dict / TorchScript export format) from the master weights."""
        self.policy.unpack(self.master[:self.n_body], self.actor_critic)
        with torch.no_grad():
            self.actor_critic.std.copy_(self.master[self.n_body:self.n_body + self.n_std])
        return self.actor_critic

    def load_module(self):
        """Adopt weights that were loaded into the nn.Module (resume)."""
        self.policy.pack(self.actor_critic, self.master[:self.n_body])
        with torch.no_grad():
            self.master[self.n_body:self.n_body + self.n_std].copy_(self.actor_critic.std)
        self._push_weights()

    @property
    def learning_rate(self):
        return float(self._lr)

    def init_storage(self, num_envs, num_transitions_per_env, actor_obs_shape, privileged_obs_shape, obs_history_shape,
                     action_shape, sliding_history=False):
        """`sliding_history`: the caller guarantees that the `obs_history` it passes to act() is a sliding window over the
        `obs` it passes (HistoryWrapper: each step drops the oldest entry and appends the new observation,
        history_wrapper.py:23), and that it is not rewritten in the middle of a rollout.  On the fused GPU path the
        storage then keeps every observation once (RolloutStorage ring) instead of every window."""
        no = int(actor_obs_shape[0])
        ring = bool(sliding_history and self.fused and PPO_Args.history_ring and no % 2 == 0 and int(obs_history_shape[0]) % no == 0)
        self.storage = RolloutStorage(num_envs, num_transitions_per_env, actor_obs_shape, privileged_obs_shape,
                                      obs_history_shape, action_shape, self.device,
                                      history_dtype=self.body.dtype, history_pad_to=64, augment=True, ring=ring)
        assert self.storage.padded_width == self.policy.Kp
        self._last_hist = torch.zeros(num_envs, self.policy.Kp, device=self.device, dtype=self.body.dtype)
        self._last_hist[:, self.storage.history_width] = 1.0
        self._X_roll = torch.zeros_like(self._last_hist) if ring else None
        if self.fused:
            from go1_gym_learn.ppo_cse.fused import FusedNet
            self._roll_net = FusedNet(self.policy, self.body, None, num_envs, self._fused_lib, with_grad=False, two_streams=False)

    def test_mode(self):
        self.actor_critic.test()

    def train_mode(self):
        self.actor_critic.train()

    # ---- rollout -----------------------------------------------------------------------------------------
    def act(self, obs, privileged_obs, obs_history):
        t = self.transition
        if self.storage.ring:
            return self._act_ring(obs, privileged_obs, obs_history)
        # the env's obs_history is a live view of its ring buffer: take the storage copy now, before env.step
        # (it is also the dtype / padding the policy GEMMs want)
        slot = self.storage.observation_histories[self.storage.step]
        self.storage.write_history(slot, obs_history, privileged_obs)
        t.observation_histories = slot
        if self._roll_net is not None and slot.shape[0] == self._roll_net.M:
            return self._act_fused(slot, obs, privileged_obs)
        mean, value = self._infer(slot)
        std = self.std.detach()
        t.actions = mean + std * torch.randn_like(mean)
        t.values = value.detach()
        t.actions_log_prob = gaussian_log_prob(t.actions, mean, std)
        t.action_mean = mean
        t.action_sigma = std.expand_as(mean)
        self._snapshot_obs(obs, privileged_obs)
        return t.actions

    def _snapshot_obs(self, obs, privileged_obs):
        """`obs` / `privileged_obs` are the environment's own output buffers, which `env.step` overwrites IN PLACE (the
        reference allocates fresh tensors every step, legged_robot.py:320-338, so holding a reference was safe there):
        the transition must keep the values the policy acted on, so they go into the storage slot now, before the step."""
        t, st = self.transition, self.storage
        s = st.step
        if s >= st.num_transitions_per_env:
            raise AssertionError("Rollout buffer overflow")
        st.observations[s].copy_(obs)
        st.privileged_observations[s].copy_(privileged_obs)
        t.observations = t.critic_observations = st.observations[s]
        t.privileged_observations = st.privileged_observations[s]

    def _act_ring(self, obs, privileged_obs, obs_history):
        """ring storage: one kernel appends the observation, assembles the inference rows and stores obs / privileged obs."""
        from go1_gym_learn.ppo_cse import fused
        st = self.storage
        s = st.step
        if s >= st.num_transitions_per_env:
            raise AssertionError("Rollout buffer overflow")
        if st.num_envs != self._roll_net.M:
            raise AssertionError("ring storage needs the static-buffer inference engine of its own batch size")
        fused.ring_step(self._fused_lib, st, s, obs.contiguous(), privileged_obs.contiguous(), obs_history, self._X_roll)
        self.transition.observation_histories = self._X_roll
        return self._act_fused(self._X_roll, obs, privileged_obs, stored=True)

    def _act_fused(self, slot, obs, privileged_obs, stored=False):
        """inference on the static-buffer engine, then ONE kernel that samples, evaluates the log-prob and writes
        actions / mean / sigma / value / log-prob straight into the storage slot (fused.act)."""
        from go1_gym_learn.ppo_cse import fused
        t, st = self.transition, self.storage
        s = st.step
        if s >= st.num_transitions_per_env:
            raise AssertionError("Rollout buffer overflow")
        with torch.no_grad():
            mean, value, _ = self._roll_net.forward(slot)
            # the sampling noise of a whole rollout as ONE generator launch at its first step (24 launches of 5 us each sat on the
            # rollout's critical path, profiles/r05b_rollout_step_timeline.txt): N(0, 1) draws either way; the fp32 / CPU path above keeps
            # torch.randn_like per step, which is what the reference-fixture tests pin
            T = st.num_transitions_per_env
            if s == 0 or self._roll_noise is None or tuple(self._roll_noise.shape) != (T, slot.shape[0], self.n_std):
                self._roll_noise = torch.randn(T, slot.shape[0], self.n_std, device=slot.device)
            noise = self._roll_noise[s]
            fused.act(self._fused_lib, mean, value, self.std, noise, st, s)
        t.actions, t.values, t.actions_log_prob = st.actions[s], st.values[s], st.actions_log_prob[s]
        t.action_mean, t.action_sigma = st.mu[s], st.sigma[s]
        if stored:          # (ring_step wrote the two observation blocks)
            t.observations = t.critic_observations = st.observations[s]
            t.privileged_observations = st.privileged_observations[s]
        else:
            self._snapshot_obs(obs, privileged_obs)
        return t.actions

    def _infer(self, rows):
        """fp32 (mean, value) of a batch of augmented history rows, no autograd."""
        with torch.no_grad():
            if self._roll_net is not None and rows.shape[0] == self._roll_net.M:
                mean, value, _ = self._roll_net.forward(rows)
                return mean[:, :self.n_std].float(), value[:, :1].float()
            mean, value, _ = self.policy.forward(self.body, rows)
            return mean.float(), value.float()

    def process_env_step(self, rewards, dones, infos):
        t = self.transition
        if t.actions is not None and self._roll_net is not None and t.actions.data_ptr() == self.storage.actions[self.storage.step].data_ptr():
            return self._process_env_step_fused(rewards, dones, infos)
        t.rewards = rewards.clone()
        t.dones = dones
        t.env_bins = infos["env_bins"]
        if 'time_outs' in infos:      # bootstrap on time-outs (reference ppo.py:84-86)
            t.rewards += PPO_Args.gamma * torch.squeeze(t.values * infos['time_outs'].unsqueeze(1).to(self.device), 1)
        self.storage.add_transitions(t)
        t.clear()
        self.actor_critic.reset(dones)

    def _process_env_step_fused(self, rewards, dones, infos):
        """the policy outputs already sit in the slot (fused.act); rewards (+ time-out bootstrap), dones and bins go
        in with one kernel, the two observation blocks with one copy each."""
        from go1_gym_learn.ppo_cse import fused
        t, st = self.transition, self.storage
        s = st.step
        tos = infos['time_outs'] if 'time_outs' in infos else None
        if tos is not None and tos.dtype == torch.bool:
            tos = tos.view(torch.uint8)
        bins = infos["env_bins"][:rewards.shape[0]]
        fused.store_step(self._fused_lib, st, s, rewards.contiguous(), dones if dones.dtype == torch.uint8 else dones.to(torch.uint8), tos,
                         bins if bins.dtype == torch.int32 else bins.to(torch.int32), PPO_Args.gamma)
        st.step += 1          # the two observation blocks were stored by act(), before the environment stepped
        t.clear()
        self.actor_critic.reset(dones)

    def compute_returns(self, last_critic_obs, last_critic_privileged_obs):
        self.storage.write_history(self._last_hist, last_critic_obs, last_critic_privileged_obs)
        _, last_values = self._infer(self._last_hist)
        self.storage.compute_returns(last_values.clone(), PPO_Args.gamma, PPO_Args.lam,
                                     fused_lib=self._fused_lib if self.fused else None)

    # ---- update ------------------------------------------------------------------------------------------
    def _adapt_lr(self, kl_mean):
        lr = self._lr
        dk = PPO_Args.desired_kl
        down = torch.clamp(lr / 1.5, min=1e-5)
        up = torch.clamp(lr * 1.5, max=1e-2)
        new = torch.where(kl_mean > dk * 2.0, down, torch.where((kl_mean < dk / 2.0) & (kl_mean > 0.0), up, lr))
        self._lr.copy_(new)
        for group in self.optimizer.param_groups:
            if torch.is_tensor(group['lr']):
                group['lr'].copy_(new)
            else:
                group['lr'] = float(new)

    # ---- one mini-batch = four stages; each stage is capture-safe (static buffers, no host reads) ------------
    def _gather(self, idx):
        st = self.storage
        f = lambda t: t.flatten(0, 1)[idx]
        return dict(hist=f(st.observation_histories), priv=f(st.privileged_observations), actions=f(st.actions),
                    values=f(st.values), adv=f(st.advantages), returns=f(st.returns), logp=f(st.actions_log_prob),
                    mu=f(st.mu), sigma=f(st.sigma))

    def _stage_ppo_backward_fused(self, idx):
        net, n = self._train_net, self.n_body
        with torch.no_grad():
            if not self._pregathered:         # graph mode gathers all mini-batches' rows once per update() instead
                self._gather_rows(idx, net.X)
            # (the flat gradient and the KL slot are clean here: update() clears them once, every fused optimiser step
            # clears what it has consumed)
            net.forward(net.X)
            net.ppo_loss(self.storage, idx, self.std, self.master.grad[n:n + self.n_std], PPO_Args, self._kl, self._acc)
            net.backward(net.X)
            # (the privileged-observation columns of the adaptation module's and the actor's first-layer rows — structurally zero
            #  weights — receive no gradient: masked by go1ppo_sum_partials / the batched weight-gradient launch, no fill pass)

    def _gather_rows(self, idx, out):
        """augmented history rows of the storage entries idx into the (len(idx), Kp) buffer `out`"""
        if self.storage.ring:
            from go1_gym_learn.ppo_cse import fused
            fused.ring_gather(self._fused_lib, self.storage, idx, out)
        else:
            torch.index_select(self.storage.observation_histories.flatten(0, 1), 0, idx, out=out)

    def _stage_adapt_backward_fused(self, idx):
        net = self._train_net          # net.X still holds this mini-batch's rows (gathered by the PPO stage)
        num_train = int(idx.numel() // 5 * 4)
        with torch.no_grad():
            net.forward_adaptation(net.X)
            net.adaptation_loss(self.storage, idx, num_train, PPO_Args.selective_adaptation_module_loss, self._acc)
            net.backward_adaptation(net.X)

    def _stage_ppo_backward(self, idx):
        """PPO loss forward + backward into the fp32 master gradient (reference ppo.py:112-155)."""
        if self._train_net is not None:
            return self._stage_ppo_backward_fused(idx)
        A = PPO_Args
        b = self._gather(idx)
        mu_batch, value_batch, _ = self.policy.forward(self.body, b["hist"])
        mu_batch, value_batch = mu_batch.float(), value_batch.float()
        std = self.std
        logp = gaussian_log_prob(b["actions"], mu_batch, std)
        entropy = gaussian_entropy(std)
        with torch.no_grad():     # KL(old || new), reference ppo.py:120-124
            kl = torch.sum(torch.log(std / b["sigma"] + 1.e-5)
                           + (torch.square(b["sigma"]) + torch.square(b["mu"] - mu_batch)) / (2.0 * torch.square(std)) - 0.5, axis=-1)
            self._kl.copy_(torch.mean(kl))
        ratio = torch.exp(logp - torch.squeeze(b["logp"]))
        adv = torch.squeeze(b["adv"])
        surrogate_loss = torch.max(-adv * ratio, -adv * torch.clamp(ratio, 1.0 - A.clip_param, 1.0 + A.clip_param)).mean()
        if A.use_clipped_value_loss:
            value_clipped = b["values"] + (value_batch - b["values"]).clamp(-A.clip_param, A.clip_param)
            value_loss = torch.max((value_batch - b["returns"]).pow(2), (value_clipped - b["returns"]).pow(2)).mean()
        else:
            value_loss = (b["returns"] - value_batch).pow(2).mean()
        loss = surrogate_loss + A.value_loss_coef * value_loss - A.entropy_coef * entropy
        loss.backward()
        self._pull_grads()
        self._acc[0] += value_loss.detach()
        self._acc[1] += surrogate_loss.detach()

    def _stage_ppo_step(self):
        """adaptive-KL learning rate, global-norm clip, Adam, refresh the compute copies."""
        A = PPO_Args
        sharded = self.dp and self._dp_shard is not None
        if self._opt is not None:
            adaptive = A.desired_kl is not None and A.schedule == 'adaptive'
            w = float(_world()) if self.dp else 1.0
            self._opt.step_(gscale=1.0 / w, max_norm=A.max_grad_norm, kl=self._kl if adaptive else None, kl_scale=1.0 / w,
                            desired_kl=A.desired_kl if adaptive else 0.0, zero_grad=True, zero_slot=self._kl,
                            between=(lambda partial: dist.all_reduce(partial)) if sharded else None,
                            pieces=self._train_net.last_pieces("ppo") if self._train_net is not None else None)
            if sharded:
                self._dp_gather_params()
            return
        g = self._g_live
        if self.dp:
            g.div_(_world())
        if A.desired_kl is not None and A.schedule == 'adaptive':
            self._adapt_lr(self._kl / _world() if self.dp else self._kl)
        self._kl.zero_()          # the slot is padding of the parameter vector: Adam must see a zero gradient there
        norm = torch.linalg.vector_norm(g)
        if sharded:               # g holds this rank's slice only: the clip needs the norm of the whole gradient
            n2 = norm * norm
            dist.all_reduce(n2)
            norm = n2.sqrt()
        g.mul_(torch.clamp(A.max_grad_norm / (norm + 1e-6), max=1.0))
        self.optimizer.step()     # (zero gradient and zero moments outside the slice: those parameters do not move)
        if sharded:
            self._dp_gather_params()
        else:
            self._push_weights()

    def _stage_adapt_backward(self, idx):
        """adaptation-module regression on the same mini-batch (reference ppo.py:163-190)."""
        if self._train_net is not None:
            return self._stage_adapt_backward_fused(idx)
        A = PPO_Args
        st = self.storage
        hist = st.observation_histories.flatten(0, 1)[idx]
        target = st.privileged_observations.flatten(0, 1)[idx]
        num_train = int(target.shape[0] // 5 * 4)
        pred = self.policy.forward(self.body, hist, want_actor=False)[2].float()
        sel = 0 if A.selective_adaptation_module_loss else slice(None)
        adaptation_loss = F.mse_loss(pred[:num_train, sel], target[:num_train, sel])
        with torch.no_grad():
            adaptation_test_loss = F.mse_loss(pred[num_train:, sel], target[num_train:, sel])
        adaptation_loss.backward()
        self._pull_grads()
        self._acc[2] += adaptation_loss.detach()
        self._acc[3] += adaptation_test_loss.detach()

    def _stage_adapt_step(self):
        if self._opt_ad is not None:
            self._opt_ad.step_(gscale=1.0 / float(_world()) if self.dp else 1.0, zero_grad=True,
                               pieces=self._train_net.last_pieces("adaptation") if self._train_net is not None else None)
            return
        if self.dp:
            self.master.grad.div_(_world())
        self.adaptation_module_optimizer.step()
        self._push_weights()

    # ---- data-parallel exchange of the PPO-stage gradient ---------------------------------------------------
    def _dp_exchange_ppo(self):
        """all-reduce of the flat gradient (+ the KL slot riding in its padding), in fp32 or — dp_grad_dtype = "bf16" —
        through a bf16 copy (every rank receives the same sum, so the replicas stay bit-identical either way); with
        dp_zero1 a reduce-scatter instead: afterwards the rank holds the summed gradient of ITS slice only (zeros elsewhere)
        and `self._dp_tail` the summed tail [std gradient | KL | pad] every rank needs."""
        g = self.master.grad
        if self._dp_shard is None:
            if self._dp_lowp is None:
                dist.all_reduce(g)
            else:
                self._dp_lowp.copy_(g)
                dist.all_reduce(self._dp_lowp)
                g.copy_(self._dp_lowp)
            return
        lo, per = self._dp_shard
        self._dp_tail.copy_(g[self.n_body:])
        dist.all_reduce(self._dp_tail)                     # std gradient + KL: a few dozen floats
        src = g
        if self._dp_lowp is not None:
            self._dp_lowp.copy_(g)
            src = self._dp_lowp
        try:
            dist.reduce_scatter_tensor(self._dp_shard_buf, src)
        except (RuntimeError, NotImplementedError):        # backend without reduce-scatter: same result from an all-reduce
            dist.all_reduce(src)
            self._dp_shard_buf.copy_(src[lo:lo + per])
        g.zero_()
        g[lo:lo + per].copy_(self._dp_shard_buf)
        self._kl.copy_(self._dp_tail[self.n_std])           # the global KL on every rank (its slot belongs to the last slice)

    def _dp_gather_params(self):
        """sharded step: every rank contributes its updated slice of the fp32 master; compute copies refreshed from it"""
        lo, per = self._dp_shard
        dist.all_gather_into_tensor(self.master.data, self.master.data[lo:lo + per].clone())
        self._push_weights()

    def _allreduce_adapt_grads(self):
        """the adaptation stage only produces gradients for the adaptation module: with the fused optimiser that is
        the leading range of the flat buffer (FlatPolicy layout; 2.3 MB instead of 13 MB over xGMI per step)."""
        for v in self._ad_grad_views:
            dist.all_reduce(v)

    def _minibatch_eager(self, idx):
        self._stage_ppo_backward(idx)
        if self.dp:
            self._dp_exchange_ppo()                    # gradient + KL slot
        self._stage_ppo_step()
        for _ in range(PPO_Args.num_adaptation_module_substeps):
            self._stage_adapt_backward(idx)
            if self.dp:
                self._allreduce_adapt_grads()
            self._stage_adapt_step()

    def _capture(self, i):
        """Record the mini-batch stages as HIP graphs (launch-bound: ~350 small kernels per mini-batch).
        Single GPU: one graph for the whole mini-batch.  Data parallel: three graphs with the RCCL all-reduces
        issued eagerly between them."""
        idx = self._idx_all[i]
        single = not self.dp and PPO_Args.num_adaptation_module_substeps == 1
        pool = None
        graphs = []
        if self.dp and PPO_Args.dp_graph_collectives and PPO_Args.num_adaptation_module_substeps == 1 and self._collectives_capturable is not False \
                and dist.get_backend() == "nccl":          # (RCCL records into a capturing stream; gloo stages through the host and cannot)
            # the whole mini-batch — stages AND collectives — as one graph.  A backend that cannot record its collectives (an RCCL build
            # without capture support) raises here and the scheme below takes over
            g, err = None, None
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    self._minibatch_eager(idx)
            except Exception as e:          # a failed plan-table copy or an out-of-memory during capture ends up here too: logged in full below
                g, err = None, e
            # every rank takes the SAME scheme: one rank replaying a graph with the collectives inside while another issues them eagerly
            # between three graphs would pair different collectives.  The outcome is agreed on here, outside any capture (MIN over the
            # ranks' success flags); a rank whose own capture succeeded discards it when another rank's did not.
            ok = torch.tensor([0 if g is None else 1], device=self.device, dtype=torch.int64)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 1:
                self._collectives_capturable = True
                return [g]
            if self._collectives_capturable is not False:
                why = f"{type(err).__name__}: {err}" if err is not None else "another rank's capture failed"
                print(f"[ppo] rank {dist.get_rank()}: one-graph capture of stages + collectives not taken ({why}); "
                      f"eager collectives between graphs on every rank", file=sys.stderr)
            self._collectives_capturable = False
            del g
            torch.cuda.synchronize()
            self.master.grad.zero_()

        def rec(fn):
            nonlocal pool
            g = torch.cuda.CUDAGraph()
            # thread-local capture mode: the RCCL watchdog thread of a data-parallel run polls events concurrently,
            # which the default (global) mode treats as a capture violation
            with torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local"):
                fn()
            pool = pool or g.pool()
            graphs.append(g)
        if single:
            rec(lambda: self._minibatch_eager(idx))
        else:
            rec(lambda: self._stage_ppo_backward(idx))
            if self._dp_shard is not None:      # the sharded step has collectives in its middle: it stays eager
                rec(lambda: self._stage_adapt_backward(idx))
            else:
                rec(lambda: (self._stage_ppo_step(), self._stage_adapt_backward(idx)))
            rec(self._stage_adapt_step)
        return graphs

    def _minibatch_replay(self, i):
        g = self._graphs[i]
        if len(g) == 1:
            g[0].replay()
            return
        g[0].replay()
        self._dp_exchange_ppo()                    # gradient + KL slot
        if self._dp_shard is not None:
            self._stage_ppo_step()
        g[1].replay()
        self._allreduce_adapt_grads()
        g[2].replay()

    def update(self):
        A = PPO_Args
        st = self.storage
        batch_size = st.num_envs * st.num_transitions_per_env
        mb = batch_size // A.num_mini_batches
        nmb = A.num_mini_batches
        if self._idx_all is None or tuple(self._idx_all.shape) != (nmb, mb):
            self._idx_all = torch.zeros(nmb, mb, dtype=torch.long, device=self.device)
            self._graphs = {}
            if self.fused:
                from go1_gym_learn.ppo_cse.fused import FusedNet
                self._train_net = FusedNet(self.policy, self.body, self.master.grad[:self.n_body], mb, self._fused_lib, with_grad=True)
                # single process: the weight-gradient slabs of a backward pass are summed by the optimiser's norm pass (no reduction
                # launch); data parallel: the all-reduce between the two needs the finished gradient
                self._train_net.defer_grad_sum = not self.dp
                # the optimiser step that changes actor.1 / critic.1 also rewrites their K-contiguous copies (the input-gradient GEMM's
                # operand): no transpose-copy launches in the backward pass.  Not with the sharded step, whose ranks step slices only.
                tr = self._train_net.adam_transposes()
                if tr and self._dp_shard is None:
                    self._train_net.refresh_transposes()
                    self._opt.set_transposes(tr)
                    self._train_net._wt_by_optimizer = True
                # one row block per mini-batch: the same nmb index sets are visited in every epoch, so in graph mode
                # their history rows are gathered once per update() (nmb gathers instead of epochs x nmb)
                self._Xall = torch.zeros(nmb, mb, self.policy.Kp, device=self.device, dtype=self.body.dtype)
        self._acc.zero_()
        if self.fused:
            self.master.grad.zero_()          # once per update; afterwards the fused optimiser steps keep it clean (and the KL slot)
        indices = torch.randperm(nmb * mb, requires_grad=False, device=self.device)   # rollout_storage.py:103
        self._idx_all.copy_(indices.view(nmb, mb))
        use_graphs = bool(self.on_gpu and A.use_hip_graphs and A.num_adaptation_module_substeps == 1
                          and (self.fused or A.use_hip_graphs == "all"))
        graph_mode = use_graphs and self._updates_done >= 1
        self._pregathered = bool(graph_mode and self.fused)
        if self._pregathered:
            for i in range(nmb):
                self._gather_rows(self._idx_all[i], self._Xall[i])
        # single process, fused update: ALL epochs x mini-batches of the update as ONE graph (round 6: every replay of a per-mini-batch graph
        # cost 9 us of idle device in front of its first kernel — 20 replays per update, profiles/r05b_minibatch_timeline.txt).  The mini-batch
        # index sets, their pre-gathered row blocks and every buffer the stages touch are static, so the 20 steps record back to back.
        # GO1_UPDATE_GRAPH=slot: one graph per mini-batch slot, replayed once per epoch (the scheme data-parallel runs keep).
        if graph_mode and self.fused and not self.dp and self._update_graph_ok and os.environ.get("GO1_UPDATE_GRAPH", "all") == "all":
            key_all = ("all", int(A.num_learning_epochs))
            g = self._graphs.get(key_all)
            if g is None:
                try:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        for epoch in range(A.num_learning_epochs):
                            for i in range(nmb):
                                self._train_net.X = self._Xall[i]
                                self._minibatch_eager(self._idx_all[i])
                    self._graphs[key_all] = g
                except Exception as err:          # an optimisation: the per-slot graphs below take over
                    print(f"[ppo] whole-update graph capture failed ({type(err).__name__}: {err}); one graph per mini-batch slot", file=sys.stderr)
                    self._update_graph_ok, g = False, None
                    torch.cuda.synchronize()
                    self.master.grad.zero_()
                    self._acc.zero_()
            if g is not None:
                g.replay()
                return self._finish_update()
        for epoch in range(A.num_learning_epochs):
            for i in range(nmb):
                idx = self._idx_all[i]
                if not self.on_gpu:
                    # the reference's update() calls actor_critic.act(...) for every mini-batch (ppo.py:107): the sample is thrown
                    # away but advances torch's generator.  Same consumption here, so that a CPU run follows the reference's
                    # random stream over any number of iterations (tests/golden/runner_iteration.npz); not on the GPU path.
                    torch.randn(mb, self.n_std)
                if self.fused:
                    self._train_net.X = self._Xall[i]
                if graph_mode:
                    if i not in self._graphs:
                        try:
                            self._graphs[i] = self._capture(i)
                        except Exception as err:      # capture is an optimisation: fall back to eager launches
                            print(f"[ppo] HIP graph capture failed ({type(err).__name__}: {err}); running eagerly", file=sys.stderr)
                            PPO_Args.use_hip_graphs = False
                            graph_mode = self._pregathered = False
                            torch.cuda.synchronize()
                            self.master.grad.zero_()
                            self._minibatch_eager(idx)
                            continue
                    self._minibatch_replay(i)
                else:
                    self._minibatch_eager(idx)
        return self._finish_update()

    def _finish_update(self):
        A = PPO_Args
        self._updates_done += 1
        num_updates = A.num_learning_epochs * A.num_mini_batches
        v, s, a, at = (self._acc / num_updates).tolist()         # the only host read of the update
        a /= A.num_adaptation_module_substeps
        at /= A.num_adaptation_module_substeps
        self.storage.clear()
        return v, s, a, 0.0, 0.0, at, 0.0, 0.0
