"""Reward container (surface of reference go1_gym/envs/rewards/corl_rewards.py:7-13).

On this stack every `_reward_*` term is evaluated inside the fused HIP step kernel
(walk-these-ways_amd/csrc/go1sim.hip `reward_term`, one case per reference function, ids in
include/go1sim.h `Go1RewardId`).  This class only records which terms exist so that
`LeggedRobot._prepare_reward_function` can warn about unknown names exactly like the reference."""
import go1sim_abi as abi


class CoRLRewards:
    names = tuple(abi.REWARD_IDS)

    def __init__(self, env):
        self.env = env

    def load_env(self, env):
        self.env = env

    def has(self, name):
        return name in abi.REWARD_IDS
