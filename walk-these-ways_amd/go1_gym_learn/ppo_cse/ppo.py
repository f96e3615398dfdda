"""PPO with adaptive-KL learning rate, clipped value loss and an adaptation-module regression step
(mirror of reference go1_gym_learn/ppo_cse/ppo.py:13-205), PyTorch-ROCm.

MI355X-first changes that do not alter the algorithm:
  * no host synchronisation inside `update()`: the KL-adaptive learning rate lives in a device tensor (Adam with a
    tensor `lr`), loss statistics are accumulated on device and read once per update;
  * bf16 autocast for the three MLPs when `PPO_Args.autocast_bf16` (fp32 master weights / Adam state);
  * data-parallel mode (only when the environments are partitioned per GPU): the flat gradient is all-reduced
    over RCCL before clipping, the KL mean is all-reduced so every rank takes the same LR branch.
"""
import torch
import torch.distributed as dist
import torch.nn as nn
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
    autocast_bf16 = False           # BASELINE config 2: "bf16 policy"
    data_parallel = True            # all-reduce gradients when torch.distributed is initialised with world_size > 1


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class PPO:
    actor_critic: ActorCritic

    def __init__(self, actor_critic, device='cpu'):
        self.device = device
        self.actor_critic = actor_critic
        self.actor_critic.to(device)
        self.actor_critic.autocast_dtype = torch.bfloat16 if PPO_Args.autocast_bf16 else None
        self.storage = None
        self.on_gpu = torch.device(device).type == "cuda"
        kw = dict(fused=True) if self.on_gpu else {}
        lr = torch.tensor(PPO_Args.learning_rate, device=device) if self.on_gpu else PPO_Args.learning_rate
        self.optimizer = optim.Adam(self.actor_critic.parameters(), lr=lr, **kw)
        self.adaptation_module_optimizer = optim.Adam(self.actor_critic.parameters(),
                                                      lr=PPO_Args.adaptation_module_learning_rate, **kw)
        if self.actor_critic.decoder:
            self.decoder_optimizer = optim.Adam(self.actor_critic.parameters(), lr=PPO_Args.adaptation_module_learning_rate)
        self.transition = RolloutStorage.Transition()
        self._lr = torch.tensor(PPO_Args.learning_rate, device=device)
        self.dp = PPO_Args.data_parallel and _world() > 1
        if self.dp:                              # identical initial weights on every rank
            for p in self.actor_critic.parameters():
                dist.broadcast(p.data, src=0)

    @property
    def learning_rate(self):
        return float(self._lr)

    def init_storage(self, num_envs, num_transitions_per_env, actor_obs_shape, privileged_obs_shape, obs_history_shape,
                     action_shape):
        bf16 = PPO_Args.autocast_bf16 and self.on_gpu
        self.storage = RolloutStorage(num_envs, num_transitions_per_env, actor_obs_shape, privileged_obs_shape,
                                      obs_history_shape, action_shape, self.device,
                                      history_dtype=torch.bfloat16 if bf16 else torch.float32,
                                      history_pad_to=8 if bf16 else 1)

    def test_mode(self):
        self.actor_critic.test()

    def train_mode(self):
        self.actor_critic.train()

    def act(self, obs, privileged_obs, obs_history):
        t = self.transition
        # the env's obs_history is a live view of its ring buffer: take the storage copy now, before env.step
        # (it is also the dtype / padding the policy GEMMs want)
        slot = self.storage.observation_histories[self.storage.step]
        slot[:, :obs_history.shape[-1]].copy_(obs_history)
        t.observation_histories = slot
        ac = self.actor_critic
        mean, value, _ = ac.fused_forward(slot, privileged_obs)
        ac.set_distribution(mean.detach())
        t.actions = ac.distribution.sample()
        t.values = value.detach()
        t.actions_log_prob = ac.get_actions_log_prob(t.actions).detach()
        t.action_mean = ac.action_mean.detach()
        t.action_sigma = ac.action_std.detach()
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
        last_values = self.actor_critic.evaluate(last_critic_obs, last_critic_privileged_obs).detach()
        if last_values.is_inference():
            last_values = last_values.clone()
        self.storage.compute_returns(last_values, PPO_Args.gamma, PPO_Args.lam)

    # ---- data parallel helpers ---------------------------------------------------------------------
    def _allreduce_grads(self, params):
        grads = [p.grad for p in params if p.grad is not None]
        if not grads:
            return
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat)
        flat.div_(_world())
        off = 0
        for g in grads:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n

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

    def update(self):
        A = PPO_Args
        dev = self.device
        acc = torch.zeros(4, device=dev)        # value, surrogate, adaptation, adaptation-test
        params = [p for p in self.actor_critic.parameters()]
        generator = self.storage.mini_batch_generator(A.num_mini_batches, A.num_learning_epochs)
        for (obs_batch, critic_obs_batch, privileged_obs_batch, obs_history_batch, actions_batch, target_values_batch,
             advantages_batch, returns_batch, old_actions_log_prob_batch, old_mu_batch, old_sigma_batch, masks_batch,
             env_bins_batch) in generator:
            mean, value_batch, _ = self.actor_critic.fused_forward(obs_history_batch, privileged_obs_batch)
            self.actor_critic.set_distribution(mean)
            actions_log_prob_batch = self.actor_critic.get_actions_log_prob(actions_batch)
            mu_batch = self.actor_critic.action_mean
            sigma_batch = self.actor_critic.action_std
            entropy_batch = self.actor_critic.entropy

            if A.desired_kl is not None and A.schedule == 'adaptive':
                with torch.no_grad():
                    kl = torch.sum(torch.log(sigma_batch / old_sigma_batch + 1.e-5)
                                   + (torch.square(old_sigma_batch) + torch.square(old_mu_batch - mu_batch))
                                   / (2.0 * torch.square(sigma_batch)) - 0.5, axis=-1)
                    self._adapt_lr(torch.mean(kl))

            ratio = torch.exp(actions_log_prob_batch - torch.squeeze(old_actions_log_prob_batch))
            adv = torch.squeeze(advantages_batch)
            surrogate_loss = torch.max(-adv * ratio, -adv * torch.clamp(ratio, 1.0 - A.clip_param, 1.0 + A.clip_param)).mean()
            if A.use_clipped_value_loss:
                value_clipped = target_values_batch + (value_batch - target_values_batch).clamp(-A.clip_param, A.clip_param)
                value_loss = torch.max((value_batch - returns_batch).pow(2), (value_clipped - returns_batch).pow(2)).mean()
            else:
                value_loss = (returns_batch - value_batch).pow(2).mean()
            loss = surrogate_loss + A.value_loss_coef * value_loss - A.entropy_coef * entropy_batch.mean()

            self.optimizer.zero_grad()
            loss.backward()
            if self.dp:
                self._allreduce_grads(params)
            nn.utils.clip_grad_norm_(self.actor_critic.parameters(), A.max_grad_norm)
            self.optimizer.step()
            acc[0] += value_loss.detach()
            acc[1] += surrogate_loss.detach()

            num_train = int(privileged_obs_batch.shape[0] // 5 * 4)
            for _ in range(A.num_adaptation_module_substeps):
                adaptation_pred = self.actor_critic.latent_padded(obs_history_batch)
                adaptation_target = privileged_obs_batch.detach()
                sel = 0 if A.selective_adaptation_module_loss else slice(None)
                adaptation_loss = F.mse_loss(adaptation_pred[:num_train, sel], adaptation_target[:num_train, sel])
                with torch.no_grad():
                    adaptation_test_loss = F.mse_loss(adaptation_pred[num_train:, sel], adaptation_target[num_train:, sel])
                self.adaptation_module_optimizer.zero_grad()
                adaptation_loss.backward()
                if self.dp:
                    self._allreduce_grads(params)
                self.adaptation_module_optimizer.step()
                acc[2] += adaptation_loss.detach()
                acc[3] += adaptation_test_loss.detach()

        num_updates = A.num_learning_epochs * A.num_mini_batches
        v, s, a, at = (acc / num_updates).tolist()         # the only host read of the update
        a /= A.num_adaptation_module_substeps
        at /= A.num_adaptation_module_substeps
        self.storage.clear()
        return v, s, a, 0.0, 0.0, at, 0.0, 0.0
