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
import copy
import math

import torch
import torch.distributed as dist
import torch.nn.functional as F
import torch.optim as optim
from params_proto import PrefixProto

from go1_gym_learn.ppo_cse import ActorCritic
from go1_gym_learn.ppo_cse import RolloutStorage
from go1_gym_learn.ppo_cse import caches  # noqa: F401


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


_HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)


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
        self.actor_critic = actor_critic
        self.actor_critic.to(device)
        self.on_gpu = torch.device(device).type == "cuda"
        self.bf16 = bool(PPO_Args.autocast_bf16 and self.on_gpu)
        self.storage = None
        ac = self.actor_critic
        modules = (ac.adaptation_module, ac.actor_body, ac.critic_body)
        # parameter order of the flat buffers: [std | adaptation module | actor | critic]
        self.master_params = [ac.std] + [p for m in modules for p in m.parameters()]
        assert len(self.master_params) == len(list(ac.parameters()))
        self.flat_param = self._flatten([p.data for p in self.master_params], torch.float32)
        self.flat_grad = torch.zeros_like(self.flat_param)      # every master .grad is a view into it
        self._bind(self.master_params, self.flat_param, self.flat_grad)
        self.adapt_params = list(ac.adaptation_module.parameters())
        offs, total = self._offsets(self.master_params)
        n_std = offs[1]                                       # padded length of the std block
        self.adapt_slice = slice(n_std, offs[1 + len(self.adapt_params)])
        self.body_slice = slice(n_std, total)
        if self.bf16:
            # bf16 compute replica of the three MLPs; `std` stays the shared fp32 parameter
            self.compute_ac = copy.deepcopy(ac)
            self.compute_ac.std = ac.std
            cmods = (self.compute_ac.adaptation_module, self.compute_ac.actor_body, self.compute_ac.critic_body)
            cparams = [p for m in cmods for p in m.parameters()]
            self.cflat_param = self._flatten([p.data for p in cparams], torch.bfloat16)
            self.cflat_grad = torch.zeros_like(self.cflat_param)
            self._bind(cparams, self.cflat_param, self.cflat_grad)
        else:
            self.compute_ac = ac
        kw = dict(fused=True) if self.on_gpu else {}
        lr = torch.tensor(PPO_Args.learning_rate, device=device) if self.on_gpu else PPO_Args.learning_rate
        self.optimizer = optim.Adam(self.master_params, lr=lr, **kw)
        # the reference builds this optimiser over all parameters (ppo.py:45-46) but only the adaptation module ever
        # receives a non-zero gradient from the adaptation loss, so only those parameters move
        self.adaptation_module_optimizer = optim.Adam(self.adapt_params, lr=PPO_Args.adaptation_module_learning_rate, **kw)
        self.transition = RolloutStorage.Transition()
        self._lr = torch.tensor(PPO_Args.learning_rate, device=device)
        self.dp = PPO_Args.data_parallel and _world() > 1
        if self.dp:                              # identical initial weights on every rank
            dist.broadcast(self.flat_param, src=0)
        self._push_weights()

    # ---- precision plumbing --------------------------------------------------------------------------
    ALIGN = 16          # elements: every parameter starts on a 32-byte (bf16) / 64-byte (fp32) boundary

    @classmethod
    def _offsets(cls, tensors):
        offs, off = [], 0
        for t in tensors:
            offs.append(off)
            off += -(-t.numel() // cls.ALIGN) * cls.ALIGN
        return offs, off

    @classmethod
    def _flatten(cls, tensors, dtype):
        offs, total = cls._offsets(tensors)
        flat = torch.zeros(total, dtype=dtype, device=tensors[0].device)
        for t, o in zip(tensors, offs):
            flat[o:o + t.numel()].copy_(t.detach().reshape(-1))
        return flat

    @classmethod
    def _bind(cls, params, flat_param, flat_grad):
        """Re-seat every parameter (and its .grad) as a view into the flat buffers."""
        offs, _ = cls._offsets(params)
        for p, off in zip(params, offs):
            n = p.numel()
            p.data = flat_param[off:off + n].view_as(p)
            p.grad = flat_grad[off:off + n].view_as(p)

    def _push_weights(self, adapt_only=False):
        """fp32 master -> bf16 compute replica: one cast kernel."""
        if self.bf16:
            sl = self.adapt_slice if adapt_only else self.body_slice
            n0 = self.body_slice.start
            self.cflat_param[sl.start - n0:sl.stop - n0].copy_(self.flat_param[sl])

    def _pull_grads(self, adapt_only=False):
        """bf16 replica gradients -> flat fp32 master gradient buffer: one cast kernel (no-op in fp32 mode)."""
        if self.bf16:
            sl = self.adapt_slice if adapt_only else self.body_slice
            n0 = self.body_slice.start
            self.flat_grad[sl].copy_(self.cflat_grad[sl.start - n0:sl.stop - n0])

    def _zero_grads(self, adapt_only=False):
        sl = self.adapt_slice if adapt_only else slice(0, self.flat_grad.numel())
        self.flat_grad[sl].zero_()
        if self.bf16:
            n0 = self.body_slice.start
            lo, hi = (sl.start - n0, sl.stop - n0) if adapt_only else (0, self.cflat_grad.numel())
            self.cflat_grad[lo:hi].zero_()

    @property
    def learning_rate(self):
        return float(self._lr)

    def init_storage(self, num_envs, num_transitions_per_env, actor_obs_shape, privileged_obs_shape, obs_history_shape,
                     action_shape):
        self.storage = RolloutStorage(num_envs, num_transitions_per_env, actor_obs_shape, privileged_obs_shape,
                                      obs_history_shape, action_shape, self.device,
                                      history_dtype=torch.bfloat16 if self.bf16 else torch.float32,
                                      history_pad_to=8 if self.bf16 else 1, augment=self.bf16)
        self.augmented = self.storage.augment
        self._last_hist = self.storage.observation_histories[0].clone()

    def test_mode(self):
        self.actor_critic.test()

    def train_mode(self):
        self.actor_critic.train()

    # ---- rollout -----------------------------------------------------------------------------------------
    def act(self, obs, privileged_obs, obs_history):
        t = self.transition
        # the env's obs_history is a live view of its ring buffer: take the storage copy now, before env.step
        # (it is also the dtype / padding the policy GEMMs want)
        slot = self.storage.observation_histories[self.storage.step]
        self.storage.write_history(slot, obs_history, privileged_obs)
        t.observation_histories = slot
        ac = self.compute_ac
        mean, value, _ = ac.fused_forward(slot, privileged_obs, augmented=self.augmented)
        mean, std = mean.detach(), ac.std.detach()
        t.actions = mean + std * torch.randn_like(mean)
        t.values = value.detach()
        t.actions_log_prob = gaussian_log_prob(t.actions, mean, std)
        t.action_mean = mean
        t.action_sigma = std.expand_as(mean)
        t.observations = obs
        t.critic_observations = obs
        t.privileged_observations = privileged_obs
        return t.actions

    def process_env_step(self, rewards, dones, infos):
        t = self.transition
        t.rewards = rewards.clone()
        t.dones = dones
        t.env_bins = infos["env_bins"]
        if 'time_outs' in infos:      # bootstrap on time-outs (reference ppo.py:84-86)
            t.rewards += PPO_Args.gamma * torch.squeeze(t.values * infos['time_outs'].unsqueeze(1).to(self.device), 1)
        self.storage.add_transitions(t)
        t.clear()
        self.actor_critic.reset(dones)

    def compute_returns(self, last_critic_obs, last_critic_privileged_obs):
        self.storage.write_history(self._last_hist, last_critic_obs, last_critic_privileged_obs)
        with torch.no_grad():
            _, last_values, _ = self.compute_ac.fused_forward(self._last_hist, last_critic_privileged_obs, augmented=self.augmented)
        self.storage.compute_returns(last_values.detach().clone(), PPO_Args.gamma, PPO_Args.lam)

    # ---- update ------------------------------------------------------------------------------------------
    def _adapt_lr(self, kl_mean):
        if self.dp:
            dist.all_reduce(kl_mean)
            kl_mean = kl_mean / _world()
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

    def _clip_and_step(self, optimizer, flat, max_norm=None):
        if self.dp:
            dist.all_reduce(flat)
            flat.div_(_world())
        if max_norm is not None:     # nn.utils.clip_grad_norm_ on the flat buffer
            flat.mul_(torch.clamp(max_norm / (torch.linalg.vector_norm(flat) + 1e-6), max=1.0))
        optimizer.step()

    def update(self):
        A = PPO_Args
        ac = self.compute_ac
        acc = torch.zeros(4, device=self.device)        # value, surrogate, adaptation, adaptation-test
        generator = self.storage.mini_batch_generator(A.num_mini_batches, A.num_learning_epochs)
        for (obs_batch, critic_obs_batch, privileged_obs_batch, obs_history_batch, actions_batch, target_values_batch,
             advantages_batch, returns_batch, old_actions_log_prob_batch, old_mu_batch, old_sigma_batch, masks_batch,
             env_bins_batch) in generator:
            self._zero_grads()
            mu_batch, value_batch, _ = ac.fused_forward(obs_history_batch, privileged_obs_batch, augmented=self.augmented)
            std = ac.std
            actions_log_prob_batch = gaussian_log_prob(actions_batch, mu_batch, std)
            entropy = gaussian_entropy(std)

            if A.desired_kl is not None and A.schedule == 'adaptive':
                with torch.no_grad():     # reference ppo.py:120-124
                    kl = torch.sum(torch.log(std / old_sigma_batch + 1.e-5)
                                   + (torch.square(old_sigma_batch) + torch.square(old_mu_batch - mu_batch))
                                   / (2.0 * torch.square(std)) - 0.5, axis=-1)
                    self._adapt_lr(torch.mean(kl))

            ratio = torch.exp(actions_log_prob_batch - torch.squeeze(old_actions_log_prob_batch))
            adv = torch.squeeze(advantages_batch)
            surrogate_loss = torch.max(-adv * ratio, -adv * torch.clamp(ratio, 1.0 - A.clip_param, 1.0 + A.clip_param)).mean()
            if A.use_clipped_value_loss:
                value_clipped = target_values_batch + (value_batch - target_values_batch).clamp(-A.clip_param, A.clip_param)
                value_loss = torch.max((value_batch - returns_batch).pow(2), (value_clipped - returns_batch).pow(2)).mean()
            else:
                value_loss = (returns_batch - value_batch).pow(2).mean()
            loss = surrogate_loss + A.value_loss_coef * value_loss - A.entropy_coef * entropy

            loss.backward()
            self._pull_grads()
            self._clip_and_step(self.optimizer, self.flat_grad, A.max_grad_norm)
            self._push_weights()
            acc[0] += value_loss.detach()
            acc[1] += surrogate_loss.detach()

            num_train = int(privileged_obs_batch.shape[0] // 5 * 4)
            for _ in range(A.num_adaptation_module_substeps):
                self._zero_grads(adapt_only=True)
                adaptation_pred = ac.latent_padded(obs_history_batch, augmented=self.augmented)
                adaptation_target = privileged_obs_batch.detach()
                sel = 0 if A.selective_adaptation_module_loss else slice(None)
                adaptation_loss = F.mse_loss(adaptation_pred[:num_train, sel], adaptation_target[:num_train, sel])
                with torch.no_grad():
                    adaptation_test_loss = F.mse_loss(adaptation_pred[num_train:, sel], adaptation_target[num_train:, sel])
                adaptation_loss.backward()
                self._pull_grads(adapt_only=True)
                self._clip_and_step(self.adaptation_module_optimizer, self.flat_grad[self.adapt_slice])
                self._push_weights(adapt_only=True)
                acc[2] += adaptation_loss.detach()
                acc[3] += adaptation_test_loss.detach()

        num_updates = A.num_learning_epochs * A.num_mini_batches
        v, s, a, at = (acc / num_updates).tolist()         # the only host read of the update
        a /= A.num_adaptation_module_substeps
        at /= A.num_adaptation_module_substeps
        self.storage.clear()
        return v, s, a, 0.0, 0.0, at, 0.0, 0.0
