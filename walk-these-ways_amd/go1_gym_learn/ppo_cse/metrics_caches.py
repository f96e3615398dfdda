"""`DistCache` / `SlotCache` under the ppo_cse path too (the reference keeps a second copy here,
go1_gym_learn/ppo_cse/metrics_caches.py; its runners import the one in go1_gym_learn.ppo, ppo_cse/__init__.py:34)."""
from go1_gym_learn.ppo.metrics_caches import DistCache, SlotCache  # noqa: F401
