"""(T, N, .) rollout buffers, GAE return scan and shuffled mini-batches (mirror of reference
go1_gym_learn/ppo_cse/rollout_storage.py:7-139).  The advantage normalisation becomes a global statistic when
the environments are sharded over ranks (one small all-reduce)."""
import torch
import torch.distributed as dist


class RolloutStorage:
    class Transition:
        _FIELDS = ("observations", "privileged_observations", "observation_histories", "critic_observations", "actions",
                   "rewards", "dones", "values", "actions_log_prob", "action_mean", "action_sigma", "env_bins")

        def __init__(self):
            for f in self._FIELDS:
                setattr(self, f, None)

        def clear(self):
            self.__init__()

    def __init__(self, num_envs, num_transitions_per_env, obs_shape, privileged_obs_shape, obs_history_shape,
                 actions_shape, device='cpu', history_dtype=torch.float32, history_pad_to=1, augment=False, ring=False):
        """`history_dtype` / `history_pad_to`: the (T, N, H*num_obs) history block is the dominant storage
        (826 MB in fp32 at N=4096); under the bf16 policy it is kept in bf16 with rows zero-padded to a multiple
        of `history_pad_to` elements, which is exactly what the policy GEMMs consume.  `augment`: the padding
        columns carry [1, privileged_obs] so that biases and the critic's privileged inputs ride in the GEMM.

        `ring`: the histories are sliding windows over one observation stream (history_wrapper.py:23), so every
        observation is stored ONCE — `obs_ring` (T + H - 1, N, num_obs), 30 MB instead of 415 MB at N=4096 — and the
        augmented rows are assembled by the kernels that consume them (include/go1ppo.h, observation ring); there is
        no `observation_histories` block then.  GPU + bf16 + augment only (the caller decides, PPO.init_storage)."""
        self.device = device
        self.obs_shape, self.privileged_obs_shape = obs_shape, privileged_obs_shape
        self.obs_history_shape, self.actions_shape = obs_history_shape, actions_shape
        T, N = num_transitions_per_env, num_envs
        z = lambda *s, **kw: torch.zeros(T, N, *s, device=self.device, **kw)
        self.observations = z(*obs_shape)
        self.privileged_observations = z(*privileged_obs_shape)
        self.history_width = int(obs_history_shape[0])
        self.augment = bool(augment)
        extra = (1 + int(privileged_obs_shape[0])) if augment else 0
        padded = -(-(self.history_width + extra) // history_pad_to) * history_pad_to
        self.padded_width = padded
        self.ring = bool(ring)
        if self.ring:
            no = int(obs_shape[0])
            assert augment and history_dtype == torch.bfloat16 and no % 2 == 0 and self.history_width % no == 0
            self.history_length = self.history_width // no
            self.obs_ring = torch.zeros(T + self.history_length - 1, N, no, device=self.device, dtype=torch.bfloat16)
            self.observation_histories = None
        else:
            self.observation_histories = z(padded, dtype=history_dtype)
            if augment:
                self.observation_histories[..., self.history_width] = 1.0
        self.rewards = z(1)
        self.actions = z(*actions_shape)
        self.dones = z(1).byte()
        self.actions_log_prob = z(1)
        self.values = z(1)
        self.returns = z(1)
        self.advantages = z(1)
        self.mu = z(*actions_shape)
        self.sigma = z(*actions_shape)
        self.env_bins = z(1)
        self.num_transitions_per_env, self.num_envs = T, N
        self.step = 0

    def write_history(self, slot, obs_history, privileged_obs):
        """copy (and cast/pad/augment) a history batch into `slot` (a (N, padded) row block of this layout)."""
        K = self.history_width
        slot[:, :K].copy_(obs_history)
        if self.augment:
            slot[:, K + 1:K + 1 + privileged_obs.shape[-1]].copy_(privileged_obs)

    def add_transitions(self, transition):
        if self.step >= self.num_transitions_per_env:
            raise AssertionError("Rollout buffer overflow")
        s = self.step
        if transition.observations.data_ptr() != self.observations[s].data_ptr():        # PPO.act stores these two itself
            self.observations[s].copy_(transition.observations)
        if transition.privileged_observations.data_ptr() != self.privileged_observations[s].data_ptr():
            self.privileged_observations[s].copy_(transition.privileged_observations)
        if self.ring:
            raise AssertionError("ring storage is filled by PPO.act (go1ppo_ring_step), not by add_transitions")
        if transition.observation_histories.data_ptr() != self.observation_histories[s].data_ptr():
            src = transition.observation_histories
            self.observation_histories[s][:, :src.shape[-1]].copy_(src)
        self.actions[s].copy_(transition.actions)
        self.rewards[s].copy_(transition.rewards.view(-1, 1))
        self.dones[s].copy_(transition.dones.view(-1, 1))
        self.values[s].copy_(transition.values)
        self.actions_log_prob[s].copy_(transition.actions_log_prob.view(-1, 1))
        self.mu[s].copy_(transition.action_mean)
        self.sigma[s].copy_(transition.action_sigma)
        self.env_bins[s].copy_(transition.env_bins.view(-1, 1))
        self.step += 1

    def clear(self):
        self.step = 0

    def compute_returns(self, last_values, gamma, lam, fused_lib=None):
        """`advantages` / `returns` are updated IN PLACE: the update's HIP graphs hold their addresses."""
        if fused_lib is not None:
            return self._compute_returns_fused(last_values, gamma, lam, fused_lib)
        advantage = 0
        for step in reversed(range(self.num_transitions_per_env)):
            next_values = last_values if step == self.num_transitions_per_env - 1 else self.values[step + 1]
            alive = 1.0 - self.dones[step].float()
            delta = self.rewards[step] + alive * gamma * next_values - self.values[step]
            advantage = delta + alive * gamma * lam * advantage
            self.returns[step] = advantage + self.values[step]
        torch.sub(self.returns, self.values, out=self.advantages)
        mean, std = self._global_mean_std(self.advantages)
        self.advantages.sub_(mean).div_(std + 1e-8)

    def _compute_returns_fused(self, last_values, gamma, lam, lib):
        """Same scan as one kernel (one thread per environment) + one normalisation kernel (csrc/go1ppo.hip)."""
        from go1_gym_learn.ppo_cse import fused
        if getattr(self, "_gae_stats", None) is None:
            self._gae_stats = torch.zeros(3, device=self.device, dtype=torch.float64)
        fused.gae(lib, self, last_values.contiguous().view(-1), gamma, lam, self._gae_stats)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self._gae_stats)          # global mean / std when the environments are sharded over ranks
        fused.normalize(lib, self, self._gae_stats)

    @staticmethod
    def _global_mean_std(x):
        """mean / unbiased std over every rank's samples (equals x.mean(), x.std() on one rank)."""
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return x.mean(), x.std()
        stats = torch.stack((x.sum(), (x * x).sum(), torch.tensor(float(x.numel()), device=x.device))).double()
        dist.all_reduce(stats)
        n = stats[2]
        mean = stats[0] / n
        var = (stats[1] - n * mean * mean) / (n - 1)
        return mean.to(x.dtype), var.clamp(min=0).sqrt().to(x.dtype)

    def get_statistics(self):
        done = self.dones
        done[-1] = 1
        flat = done.permute(1, 0, 2).reshape(-1, 1)
        idx = torch.cat((flat.new_tensor([-1], dtype=torch.int64), flat.nonzero(as_tuple=False)[:, 0]))
        return (idx[1:] - idx[:-1]).float().mean(), self.rewards.mean()

    def mini_batch_generator(self, num_mini_batches, num_epochs=8):
        batch_size = self.num_envs * self.num_transitions_per_env
        mini_batch_size = batch_size // num_mini_batches
        indices = torch.randperm(num_mini_batches * mini_batch_size, requires_grad=False, device=self.device)
        flat = lambda t: t.flatten(0, 1)
        assert not self.ring, "ring storage: mini-batch rows are assembled by go1ppo_ring_gather (PPO.update)"
        observations, privileged_obs, obs_history = flat(self.observations), flat(self.privileged_observations), flat(self.observation_histories)
        actions, values, returns = flat(self.actions), flat(self.values), flat(self.returns)
        old_log_prob, advantages = flat(self.actions_log_prob), flat(self.advantages)
        old_mu, old_sigma, old_bins = flat(self.mu), flat(self.sigma), flat(self.env_bins)
        for _ in range(num_epochs):
            for i in range(num_mini_batches):
                idx = indices[i * mini_batch_size:(i + 1) * mini_batch_size]
                yield (observations[idx], observations[idx], privileged_obs[idx], obs_history[idx], actions[idx], values[idx],
                       advantages[idx], returns[idx], old_log_prob[idx], old_mu[idx], old_sigma[idx], None, old_bins[idx])

    def reccurent_mini_batch_generator(self, num_mini_batches, num_epochs=8):
        """Mini-batches of whole trajectories for a recurrent policy (reference rollout_storage.py:141-180; name as spelt
        there).  The feed-forward ppo_cse policy never asks for it: kept for the storage's interface, block storage only."""
        from go1_gym_learn.utils import split_and_pad_trajectories
        assert not self.ring, "ring storage holds no per-step history rows"
        K = self.history_width
        obs_traj, masks = split_and_pad_trajectories(self.observations, self.dones)
        priv_traj, _ = split_and_pad_trajectories(self.privileged_observations, self.dones)
        hist_traj, _ = split_and_pad_trajectories(self.observation_histories[..., :K].float(), self.dones)
        per_batch = self.num_envs // num_mini_batches
        dones = self.dones.squeeze(-1).bool()
        starts_here = torch.ones_like(dones)                    # a trajectory starts at t = 0 and after every done
        starts_here[1:] = dones[:-1]
        for _ in range(num_epochs):
            first = 0
            for i in range(num_mini_batches):
                env = slice(i * per_batch, (i + 1) * per_batch)
                last = first + int(starts_here[:, env].sum())
                traj = slice(first, last)
                yield (obs_traj[:, traj], obs_traj[:, traj], priv_traj[:, traj], hist_traj[:, traj], self.actions[:, env],
                       self.values[:, env], self.advantages[:, env], self.returns[:, env], self.actions_log_prob[:, env],
                       self.mu[:, env], self.sigma[:, env], masks[:, traj])
                first = last
