"""The reference's go1_gym_learn/utils only serves recurrent policies (`split_and_pad_trajectories`); the
ppo_cse policy is feed-forward (actor_critic.py:20 `is_recurrent = False`), so nothing is needed here."""
