"""PPO of the older (teacher-student) runner (mirror of reference go1_gym_learn/ppo/ppo.py:16-178): clipped
surrogate + clipped value loss + entropy bonus through the privileged-latent policy, KL-adaptive learning rate,
then a regression step of the adaptation module onto the encoder's latent.  Plain PyTorch autograd (SURVEY.md §8f
rank 4 — not the hot path); the Gaussian algebra is written out instead of going through torch.distributions."""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.optim as optim
from params_proto import PrefixProto

from go1_gym_learn.ppo import ActorCritic, RolloutStorage, caches  # noqa: F401
from go1_gym_learn.ppo_cse.ppo import gaussian_entropy, gaussian_log_prob


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


class PPO:
    actor_critic: ActorCritic

    def __init__(self, actor_critic, device='cpu'):
        self.device = device
        self.actor_critic = actor_critic.to(device)
        self.storage = None
        # both optimisers see every parameter (reference :44-46); the adaptation step only produces gradients for the
        # adaptation module, and parameters without a gradient are skipped by Adam
        self.optimizer = optim.Adam(self.actor_critic.parameters(), lr=PPO_Args.learning_rate)
        self.adaptation_module_optimizer = optim.Adam(self.actor_critic.parameters(), lr=PPO_Args.adaptation_module_learning_rate)
        self.transition = RolloutStorage.Transition()
        self.learning_rate = PPO_Args.learning_rate

    def init_storage(self, num_envs, num_transitions_per_env, actor_obs_shape, privileged_obs_shape, obs_history_shape, action_shape):
        self.storage = RolloutStorage(num_envs, num_transitions_per_env, actor_obs_shape, privileged_obs_shape, obs_history_shape,
                                      action_shape, self.device)

    def test_mode(self):
        self.actor_critic.test()

    def train_mode(self):
        self.actor_critic.train()

    # ---- rollout -----------------------------------------------------------------------------------------------------
    def act(self, obs, privileged_obs, obs_history):
        ac, tr = self.actor_critic, self.transition
        tr.actions = ac.act(obs, privileged_obs).detach()
        tr.values = ac.evaluate(obs, privileged_obs).detach()
        tr.actions_log_prob = ac.get_actions_log_prob(tr.actions).detach()
        tr.action_mean = ac.action_mean.detach()
        tr.action_sigma = ac.action_std.detach()
        # obs / privileged_obs / obs_history are the environment's own buffers and env.step overwrites them in place (the
        # reference builds fresh tensors every step, legged_robot.py:320-338): the slot gets the values acted on, now
        st, s = self.storage, self.storage.step
        if s >= st.num_transitions_per_env:
            raise AssertionError("Rollout buffer overflow")
        st.observations[s].copy_(obs)
        st.privileged_observations[s].copy_(privileged_obs)
        st.observation_histories[s][:, :obs_history.shape[-1]].copy_(obs_history)
        tr.observations = tr.critic_observations = st.observations[s]
        tr.privileged_observations = st.privileged_observations[s]
        tr.observation_histories = st.observation_histories[s]
        return tr.actions

    def process_env_step(self, rewards, dones, infos):
        tr = self.transition
        tr.rewards = rewards.clone()
        tr.dones = dones
        tr.env_bins = infos["env_bins"]
        if 'time_outs' in infos:           # bootstrap the value of the cut-off tail
            tr.rewards += PPO_Args.gamma * torch.squeeze(tr.values * infos['time_outs'].unsqueeze(1).to(self.device), 1)
        self.storage.add_transitions(tr)
        tr.clear()
        self.actor_critic.reset(dones)

    def compute_returns(self, last_critic_obs, last_critic_privileged_obs):
        last_values = self.actor_critic.evaluate(last_critic_obs, last_critic_privileged_obs).detach()
        self.storage.compute_returns(last_values, PPO_Args.gamma, PPO_Args.lam)

    # ---- update ------------------------------------------------------------------------------------------------------
    def _adapt_learning_rate(self, mu, sigma, old_mu, old_sigma):
        with torch.no_grad():
            kl = (torch.log(sigma / old_sigma + 1.e-5) + (old_sigma.square() + (old_mu - mu).square()) / (2.0 * sigma.square()) - 0.5).sum(-1)
            kl_mean = float(kl.mean())
        if kl_mean > PPO_Args.desired_kl * 2.0:
            self.learning_rate = max(1e-5, self.learning_rate / 1.5)
        elif 0.0 < kl_mean < PPO_Args.desired_kl / 2.0:
            self.learning_rate = min(1e-2, self.learning_rate * 1.5)
        for group in self.optimizer.param_groups:
            group['lr'] = self.learning_rate

    def update(self):
        A, ac = PPO_Args, self.actor_critic
        sums = np.zeros(3)
        for (obs, critic_obs, priv, hist, actions, old_values, advantages, returns, old_logp, old_mu, old_sigma, _masks,
             env_bins) in self.storage.mini_batch_generator(A.num_mini_batches, A.num_learning_epochs):
            latent = ac.env_factor_encoder(priv)
            mu = ac.actor_body(torch.cat((obs, latent), dim=-1))
            value = ac.critic_body(torch.cat((critic_obs, latent), dim=-1))
            sigma = mu * 0. + ac.std
            logp = gaussian_log_prob(actions, mu, ac.std)
            entropy = gaussian_entropy(ac.std)
            if A.desired_kl is not None and A.schedule == 'adaptive':
                self._adapt_learning_rate(mu, sigma, old_mu, old_sigma)
            adv = advantages.squeeze(-1)
            ratio = torch.exp(logp - old_logp.squeeze(-1))
            surrogate_loss = torch.max(-adv * ratio, -adv * ratio.clamp(1.0 - A.clip_param, 1.0 + A.clip_param)).mean()
            if A.use_clipped_value_loss:
                clipped = old_values + (value - old_values).clamp(-A.clip_param, A.clip_param)
                value_loss = torch.max((value - returns).square(), (clipped - returns).square()).mean()
            else:
                value_loss = (returns - value).square().mean()
            loss = surrogate_loss + A.value_loss_coef * value_loss - A.entropy_coef * entropy
            self.optimizer.zero_grad()
            loss.backward()
            nn.utils.clip_grad_norm_(ac.parameters(), A.max_grad_norm)
            self.optimizer.step()
            sums[0] += value_loss.item()
            sums[1] += surrogate_loss.item()
            for _ in range(A.num_adaptation_module_substeps):
                pred = ac.adaptation_module(hist)
                with torch.no_grad():
                    target = ac.env_factor_encoder(priv)
                    residual = (target - pred).norm(dim=1)
                    caches.slot_cache.log(env_bins[:, 0].cpu().numpy().astype(np.uint8), sysid_residual=residual.cpu().numpy())
                adaptation_loss = F.mse_loss(pred, target)
                self.adaptation_module_optimizer.zero_grad()
                adaptation_loss.backward()
                self.adaptation_module_optimizer.step()
                sums[2] += adaptation_loss.item()
        n = A.num_learning_epochs * A.num_mini_batches
        self.storage.clear()
        return sums[0] / n, sums[1] / n, sums[2] / (n * A.num_adaptation_module_substeps)
