from .vec_env import VecEnv  # noqa: F401  (reference go1_gym_learn/env/__init__.py:3)
