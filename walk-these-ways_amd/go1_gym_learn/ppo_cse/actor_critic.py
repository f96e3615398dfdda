"""Actor-critic with a learned adaptation module (mirror of reference go1_gym_learn/ppo_cse/actor_critic.py:7-166).

Same module names and layer layout, so state_dict keys (`adaptation_module.N.*`, `actor_body.N.*`,
`critic_body.N.*`, `std`) and the exported TorchScript files load interchangeably with the reference
(SURVEY.md §8f rank 2).  `fused_forward` is the path PPO uses on MI355X (one GEMM over the shared 2100-wide
history input); the reference-surface methods below it keep their original semantics."""
import torch
import torch.nn as nn
import torch.nn.functional as F
from params_proto import PrefixProto
from torch.distributions import Normal


class AC_Args(PrefixProto, cli=False):
    init_noise_std = 1.0
    actor_hidden_dims = [512, 256, 128]
    critic_hidden_dims = [512, 256, 128]
    activation = 'elu'
    adaptation_module_branch_hidden_dims = [256, 128]
    use_decoder = False


_ACTIVATIONS = {"elu": nn.ELU, "selu": nn.SELU, "relu": nn.ReLU, "crelu": nn.ReLU, "lrelu": nn.LeakyReLU,
                "tanh": nn.Tanh, "sigmoid": nn.Sigmoid}


def get_activation(act_name):
    if act_name not in _ACTIVATIONS:
        print("invalid activation function!")
        return None
    return _ACTIVATIONS[act_name]()


def _mlp(sizes, act):
    layers = []
    for i in range(len(sizes) - 1):
        layers.append(nn.Linear(sizes[i], sizes[i + 1]))
        if i < len(sizes) - 2:
            layers.append(act)
    return nn.Sequential(*layers)


class ActorCritic(nn.Module):
    is_recurrent = False

    def __init__(self, num_obs, num_privileged_obs, num_obs_history, num_actions, **kwargs):
        if kwargs:
            print("ActorCritic.__init__ got unexpected arguments, which will be ignored: " + str(list(kwargs)))
        self.decoder = AC_Args.use_decoder
        super().__init__()
        self.num_obs_history = num_obs_history
        self.num_privileged_obs = num_privileged_obs
        act = get_activation(AC_Args.activation)
        self.adaptation_module = _mlp([num_obs_history] + list(AC_Args.adaptation_module_branch_hidden_dims) + [num_privileged_obs], act)
        self.actor_body = _mlp([num_privileged_obs + num_obs_history] + list(AC_Args.actor_hidden_dims) + [num_actions], act)
        self.critic_body = _mlp([num_privileged_obs + num_obs_history] + list(AC_Args.critic_hidden_dims) + [1], act)
        self.std = nn.Parameter(AC_Args.init_noise_std * torch.ones(num_actions))
        self.distribution = None
        Normal.set_default_validate_args = False

    @staticmethod
    def init_weights(sequential, scales):
        """orthogonal initialisation of the Linear layers of `sequential`, gain scales[i] for the i-th (reference
        actor_critic.py:89-93, unused there as well)"""
        linears = [m for m in sequential if isinstance(m, nn.Linear)]
        for layer, gain in zip(linears, scales):
            torch.nn.init.orthogonal_(layer.weight, gain=gain)

    # -- fused path used by PPO on MI355X ------------------------------------------------------------
    @staticmethod
    def _side(y, z, Wz, b):
        """y + z @ Wz^T + b for a side input with a tiny inner dimension (2 latent / privileged columns):
        broadcast multiply-adds instead of a K=2 GEMM, which GEMM libraries handle poorly."""
        if z.shape[-1] <= 4:
            out = y + b
            for i in range(z.shape[-1]):
                out = torch.addcmul(out, z[:, i:i + 1], Wz[:, i])
            return out
        return y + F.linear(z, Wz, b)

    @staticmethod
    def _head(seq, x, min_cols=64):
        """seq[1:] applied to first-layer pre-activations x, with the final Linear's output width padded to
        `min_cols` zero rows: GEMMs with 1 / 2 / 12 output columns (value, latent, action mean) are served by very
        slow kernels, a 64-column GEMM is not; the extra columns are dropped."""
        body, last = seq[1:-1], seq[-1]
        h = body(x)
        n = last.weight.shape[0]
        if n >= min_cols or not h.is_cuda:
            return last(h)
        W = F.pad(last.weight, (0, 0, 0, min_cols - n))
        b = F.pad(last.bias, (0, min_cols - n))
        return F.linear(h, W, b).split([n, min_cols - n], dim=1)[0]

    def fused_forward(self, observation_history, privileged_observations=None, want_value=True, augmented=False):
        """(action mean, value, latent) with ONE GEMM over the 2100-wide history for the first layers of the
        adaptation module, the actor and the critic (they share the input: 256 + 512 + 512 output columns),
        instead of three GEMMs plus two (M, 2102) concatenations.  Mathematically identical to
        act()/evaluate(): W [h ; z] = W_h h + W_z z.  Runs in the dtype of the module's weights (the bf16 compute
        replica or the fp32 master).

        `observation_history` may be the storage's padded copy (row length a multiple of 8 elements so that every
        GEMM operand is 16-byte aligned).  With `augmented=True` the padding carries [1, privileged_obs...] in
        columns K, K+1.. (written by the storage), so the first-layer biases and the critic's privileged-input
        weights ride inside the same GEMM: x' = [h, 1, p, 0], W' = [W_h, b, W_p, 0]."""
        K = self.num_obs_history
        la, lc, ld = self.actor_body[0], self.critic_body[0], self.adaptation_module[0]
        wdt, odt = ld.weight.dtype, self.std.dtype
        x = observation_history if observation_history.dtype == wdt else observation_history.to(wdt)
        pad = x.shape[-1] - K
        nd, na = ld.weight.shape[0], la.weight.shape[0]
        npv = self.num_privileged_obs
        if augmented:
            assert pad >= 1 + npv
            z = self._zeros(max(nd, na), pad, wdt, x.device)          # cached constant block of zeros
            rows = [torch.cat((ld.weight, ld.bias.unsqueeze(1), z[:nd, :pad - 1]), dim=1),
                    torch.cat((la.weight[:, :K], la.bias.unsqueeze(1), z[:na, :pad - 1]), dim=1)]
            sizes = [nd, na]
            if want_value:
                nc = lc.weight.shape[0]
                rows.append(torch.cat((lc.weight[:, :K], lc.bias.unsqueeze(1), lc.weight[:, K:], z[:nc, :pad - 1 - npv]), dim=1))
                sizes.append(nc)
            # split (not slicing): its backward is one concatenation instead of a zero-fill + add per slice
            ys = F.linear(x, torch.cat(rows, dim=0)).split(sizes, dim=1)
            latent = self._head(self.adaptation_module, ys[0])
            a1 = ys[1]
            for i in range(npv):
                a1 = torch.addcmul(a1, latent[:, i:i + 1], la.weight[:, K + i])
            mean = self._head(self.actor_body, a1).to(odt)
            value = self._head(self.critic_body, ys[2]).to(odt) if want_value else None
            return mean, value, latent.to(odt)
        parts = [ld.weight, la.weight[:, :K]] + ([lc.weight[:, :K]] if want_value else [])
        Wh = torch.cat(parts, dim=0)
        if pad:
            Wh = F.pad(Wh, (0, pad))
        y = F.linear(x, Wh)
        latent = self._head(self.adaptation_module, y[:, :nd] + ld.bias)
        mean = self._head(self.actor_body, self._side(y[:, nd:nd + na], latent, la.weight[:, K:], la.bias)).to(odt)
        value = None
        if want_value:
            p = privileged_observations.to(wdt)
            value = self._head(self.critic_body, self._side(y[:, nd + na:], p, lc.weight[:, K:], lc.bias)).to(odt)
        return mean, value, latent.to(odt)

    def _zeros(self, rows, cols, dtype, device):
        key = (rows, cols, dtype, str(device))
        cache = self.__dict__.setdefault("_zero_cache", {})
        if key not in cache:
            cache[key] = torch.zeros(rows, cols, dtype=dtype, device=device)
        return cache[key]

    def latent_padded(self, observation_history, augmented=False):
        """adaptation module on a (possibly padded / augmented) history batch."""
        K = self.num_obs_history
        ld = self.adaptation_module[0]
        x = observation_history if observation_history.dtype == ld.weight.dtype else observation_history.to(ld.weight.dtype)
        pad = x.shape[-1] - K
        if augmented:
            z = self._zeros(ld.weight.shape[0], pad, ld.weight.dtype, x.device)
            W = torch.cat((ld.weight, ld.bias.unsqueeze(1), z[:, :pad - 1]), dim=1)
            return self._head(self.adaptation_module, F.linear(x, W)).to(self.std.dtype)
        W = F.pad(ld.weight, (0, pad)) if pad else ld.weight
        return self._head(self.adaptation_module, F.linear(x, W, ld.bias)).to(self.std.dtype)

    def set_distribution(self, mean):
        self.distribution = Normal(mean, mean * 0. + self.std)

    def _latent(self, observation_history):
        return self.adaptation_module(observation_history)

    def _actor(self, observation_history, latent):
        return self.actor_body(torch.cat((observation_history, latent), dim=-1))

    # -- reference surface -----------------------------------------------------------------------------
    def reset(self, dones=None):
        pass

    def forward(self):
        raise NotImplementedError

    @property
    def action_mean(self):
        return self.distribution.mean

    @property
    def action_std(self):
        return self.distribution.stddev

    @property
    def entropy(self):
        return self.distribution.entropy().sum(dim=-1)

    def update_distribution(self, observation_history):
        mean = self._actor(observation_history, self._latent(observation_history))
        self.distribution = Normal(mean, mean * 0. + self.std)

    def act(self, observation_history, **kwargs):
        self.update_distribution(observation_history)
        return self.distribution.sample()

    def get_actions_log_prob(self, actions):
        return self.distribution.log_prob(actions).sum(dim=-1)

    def act_expert(self, ob, policy_info={}):
        return self.act_teacher(ob["obs_history"], ob["privileged_obs"])

    def act_inference(self, ob, policy_info={}):
        return self.act_student(ob["obs_history"], policy_info=policy_info)

    def act_student(self, observation_history, policy_info={}):
        latent = self._latent(observation_history)
        policy_info["latents"] = latent.detach().cpu().numpy()
        return self._actor(observation_history, latent)

    def act_teacher(self, observation_history, privileged_info, policy_info={}):
        policy_info["latents"] = privileged_info
        return self._actor(observation_history, privileged_info)

    def evaluate(self, observation_history, privileged_observations, **kwargs):
        return self.critic_body(torch.cat((observation_history, privileged_observations), dim=-1))

    def get_student_latent(self, observation_history):
        return self._latent(observation_history)
