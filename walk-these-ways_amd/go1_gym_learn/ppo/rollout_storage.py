"""The older runner's rollout storage is the same class as ppo_cse's (the reference's two files are identical:
go1_gym_learn/ppo/rollout_storage.py == go1_gym_learn/ppo_cse/rollout_storage.py)."""
from go1_gym_learn.ppo_cse.rollout_storage import RolloutStorage  # noqa: F401
