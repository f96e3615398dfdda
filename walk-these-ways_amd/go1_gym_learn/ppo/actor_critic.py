"""Teacher-student actor-critic of the older runner (mirror of reference go1_gym_learn/ppo/actor_critic.py:9-177):
an environment-factor encoder maps the privileged observation to a latent the actor and the critic consume during
training; the adaptation module regresses the same latent from the observation history for deployment.

Module names and layer layout follow the reference, so state_dict keys (`env_factor_encoder.N.*` and its alias
`encoder.N.*`, `adaptation_module.N.*`, `actor_body.N.*`, `critic_body.N.*`, `std`) load interchangeably.
Plain PyTorch: this runner is not on the MI355X hot path; it runs on ROCm as is."""
import torch
import torch.nn as nn
from params_proto import PrefixProto
from torch.distributions import Normal

from go1_gym_learn.ppo_cse.actor_critic import get_activation  # noqa: F401  (same table of activations)


class AC_Args(PrefixProto, cli=False):
    init_noise_std = 1.0
    actor_hidden_dims = [512, 256, 128]
    critic_hidden_dims = [512, 256, 128]
    activation = 'elu'
    adaptation_module_branch_hidden_dims = [[256, 32]]
    env_factor_encoder_branch_input_dims = [18]
    env_factor_encoder_branch_latent_dims = [18]
    env_factor_encoder_branch_hidden_dims = [[256, 128]]


def _chain(widths, act):
    """Linear(w0, w1), act, Linear(w1, w2), act, ..., Linear(w[-2], w[-1]) — no activation behind the last layer."""
    mods = []
    for a, b in zip(widths[:-1], widths[1:]):
        mods += [nn.Linear(a, b), act]
    return nn.Sequential(*mods[:-1])


class ActorCritic(nn.Module):
    is_recurrent = False

    def __init__(self, num_obs, num_privileged_obs, num_obs_history, num_actions, **kwargs):
        if kwargs:
            print("ActorCritic.__init__ got unexpected arguments, which will be ignored: " + str(list(kwargs)))
        super().__init__()
        act = get_activation(AC_Args.activation)
        A = AC_Args
        # the reference loops over "branches" but keeps only the last one it builds (:37-73); one branch is configured
        enc_in, enc_hidden, latent = (A.env_factor_encoder_branch_input_dims[-1], A.env_factor_encoder_branch_hidden_dims[-1],
                                      A.env_factor_encoder_branch_latent_dims[-1])
        self.env_factor_encoder = _chain([enc_in] + list(enc_hidden) + [latent], act)
        self.add_module("encoder", self.env_factor_encoder)                      # second name of the same module (:54)
        n_ad = min(len(A.adaptation_module_branch_hidden_dims), len(A.env_factor_encoder_branch_latent_dims))
        self.adaptation_module = _chain([num_obs_history] + list(A.adaptation_module_branch_hidden_dims[n_ad - 1]) +
                                        [A.env_factor_encoder_branch_latent_dims[n_ad - 1]], act)
        total_latent = int(sum(A.env_factor_encoder_branch_latent_dims))
        self.actor_body = _chain([total_latent + num_obs] + list(A.actor_hidden_dims) + [num_actions], act)
        self.critic_body = _chain([total_latent + num_obs] + list(A.critic_hidden_dims) + [1], act)
        print(f"Environment Factor Encoder: {self.env_factor_encoder}")
        print(f"Adaptation Module: {self.adaptation_module}")
        print(f"Actor MLP: {self.actor_body}")
        print(f"Critic MLP: {self.critic_body}")
        self.std = nn.Parameter(A.init_noise_std * torch.ones(num_actions))
        self.distribution = None
        Normal.set_default_validate_args = False

    @staticmethod
    def init_weights(sequential, scales):
        linears = [m for m in sequential if isinstance(m, nn.Linear)]
        for gain, lin in zip(scales, linears):
            torch.nn.init.orthogonal_(lin.weight, gain=gain)

    def reset(self, dones=None):
        pass

    def forward(self):
        raise NotImplementedError

    # ---- distribution surface --------------------------------------------------------------------------------------
    @property
    def action_mean(self):
        return self.distribution.mean

    @property
    def action_std(self):
        return self.distribution.stddev

    @property
    def entropy(self):
        return self.distribution.entropy().sum(dim=-1)

    def _teacher_mean(self, observations, privileged_observations):
        latent = self.env_factor_encoder(privileged_observations)
        return self.actor_body(torch.cat((observations, latent), dim=-1)), latent

    def update_distribution(self, observations, privileged_observations):
        mean, _ = self._teacher_mean(observations, privileged_observations)
        self.distribution = Normal(mean, mean * 0. + self.std)

    def act(self, observations, privileged_observations, **kwargs):
        self.update_distribution(observations, privileged_observations)
        return self.distribution.sample()

    def get_actions_log_prob(self, actions):
        return self.distribution.log_prob(actions).sum(dim=-1)

    def evaluate(self, critic_observations, privileged_observations, **kwargs):
        latent = self.env_factor_encoder(privileged_observations)
        return self.critic_body(torch.cat((critic_observations, latent), dim=-1))

    # ---- inference surfaces ----------------------------------------------------------------------------------------
    def act_expert(self, ob, policy_info={}):
        return self.act_teacher(ob["obs"], ob["privileged_obs"])

    def act_inference(self, ob, policy_info={}):
        if ob["privileged_obs"] is not None:
            policy_info["gt_latents"] = self.env_factor_encoder(ob["privileged_obs"]).detach().cpu().numpy()
        return self.act_student(ob["obs"], ob["obs_history"])

    def act_student(self, observations, observation_history, policy_info={}):
        latent = self.adaptation_module(observation_history)
        policy_info["latents"] = latent.detach().cpu().numpy()
        return self.actor_body(torch.cat((observations, latent), dim=-1))

    def act_teacher(self, observations, privileged_info, policy_info={}):
        mean, latent = self._teacher_mean(observations, privileged_info)
        policy_info["latents"] = latent.detach().cpu().numpy()
        return mean
