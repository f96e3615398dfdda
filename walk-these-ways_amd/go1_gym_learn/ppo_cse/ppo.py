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
        self.actor_critic = actor_critic          # public / checkpoint format; refreshed by sync_module()
        self.actor_critic.to(device)
        self.on_gpu = torch.device(device).type == "cuda"
        self.bf16 = bool(PPO_Args.autocast_bf16 and self.on_gpu)
        self.storage = None
        ac = self.actor_critic
        self.policy = FlatPolicy(ac)
        n = self.policy.numel
        self.n_body, self.n_std = n, ac.std.numel()
        # fp32 master: [flat policy | std | pad]; the single tensor both optimisers step
        self.master = torch.zeros(n + 16, device=device)
        self.policy.pack(ac, self.master[:n])
        self.master[n:n + self.n_std].copy_(ac.std.detach())
        self.master.grad = torch.zeros_like(self.master)
        pol = self.policy
        self._priv_cols_grad = pol._block(self.master.grad[:n], "W1")[:pol.first[0] + pol.first[1], pol.K + 1:pol.K + 1 + pol.npv]
        # compute copies (leaves of the autograd graph): policy body in bf16 or fp32, std always fp32
        self.body = torch.zeros(n, device=device, dtype=torch.bfloat16 if self.bf16 else torch.float32).requires_grad_()
        self.std = torch.zeros(self.n_std, device=device).requires_grad_()
        kw = dict(fused=True) if self.on_gpu else {}
        lr = torch.tensor(PPO_Args.learning_rate, device=device) if self.on_gpu else PPO_Args.learning_rate
        self.optimizer = optim.Adam([self.master], lr=lr, **kw)
        # the reference builds this optimiser over all parameters (ppo.py:45-46) but only the adaptation module ever
        # receives a non-zero gradient from the adaptation loss; with exactly-zero gradients Adam's moments stay zero
        # and the other parameters do not move, so stepping the whole flat buffer is equivalent
        self.adaptation_module_optimizer = optim.Adam([self.master], lr=PPO_Args.adaptation_module_learning_rate, **kw)
        self.transition = RolloutStorage.Transition()
        self._lr = torch.tensor(PPO_Args.learning_rate, device=device)
        self.dp = PPO_Args.data_parallel and _world() > 1
        if self.dp:                              # identical initial weights on every rank
            dist.broadcast(self.master, src=0)
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
                     action_shape):
        self.storage = RolloutStorage(num_envs, num_transitions_per_env, actor_obs_shape, privileged_obs_shape,
                                      obs_history_shape, action_shape, self.device,
                                      history_dtype=self.body.dtype, history_pad_to=8, augment=True)
        assert self.storage.observation_histories.shape[-1] == self.policy.Kp
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
        with torch.no_grad():
            mean, value, _ = self.policy.forward(self.body, slot)
        mean, value, std = mean.float(), value.float(), self.std.detach()
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
            _, last_values, _ = self.policy.forward(self.body, self._last_hist)
        self.storage.compute_returns(last_values.float().clone(), PPO_Args.gamma, PPO_Args.lam)

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
        acc = torch.zeros(4, device=self.device)        # value, surrogate, adaptation, adaptation-test
        generator = self.storage.mini_batch_generator(A.num_mini_batches, A.num_learning_epochs)
        for (obs_batch, critic_obs_batch, privileged_obs_batch, obs_history_batch, actions_batch, target_values_batch,
             advantages_batch, returns_batch, old_actions_log_prob_batch, old_mu_batch, old_sigma_batch, masks_batch,
             env_bins_batch) in generator:
            mu_batch, value_batch, _ = self.policy.forward(self.body, obs_history_batch)
            mu_batch, value_batch = mu_batch.float(), value_batch.float()
            std = self.std
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
            self._clip_and_step(self.optimizer, self.master.grad, A.max_grad_norm)
            self._push_weights()
            acc[0] += value_loss.detach()
            acc[1] += surrogate_loss.detach()

            num_train = int(privileged_obs_batch.shape[0] // 5 * 4)
            for _ in range(A.num_adaptation_module_substeps):
                adaptation_pred = self.policy.forward(self.body, obs_history_batch, want_actor=False)[2].float()
                adaptation_target = privileged_obs_batch.detach()
                sel = 0 if A.selective_adaptation_module_loss else slice(None)
                adaptation_loss = F.mse_loss(adaptation_pred[:num_train, sel], adaptation_target[:num_train, sel])
                with torch.no_grad():
                    adaptation_test_loss = F.mse_loss(adaptation_pred[num_train:, sel], adaptation_target[num_train:, sel])
                adaptation_loss.backward()
                self._pull_grads()
                self._clip_and_step(self.adaptation_module_optimizer, self.master.grad)
                self._push_weights()
                acc[2] += adaptation_loss.detach()
                acc[3] += adaptation_test_loss.detach()

        num_updates = A.num_learning_epochs * A.num_mini_batches
        v, s, a, at = (acc / num_updates).tolist()         # the only host read of the update
        a /= A.num_adaptation_module_substeps
        at /= A.num_adaptation_module_substeps
        self.storage.clear()
        return v, s, a, 0.0, 0.0, at, 0.0, 0.0
