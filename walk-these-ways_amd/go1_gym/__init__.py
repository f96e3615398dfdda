"""MI355X-native `go1_gym`: same module paths and env surface as the reference package
(reference go1_gym/__init__.py:1-4), physics and tensor maps executed by libgo1sim (HIP)."""
import os

MINI_GYM_ROOT_DIR = os.path.dirname(os.path.dirname(os.path.realpath(__file__)))
MINI_GYM_ENVS_DIR = os.path.join(MINI_GYM_ROOT_DIR, 'go1_gym', 'envs')
