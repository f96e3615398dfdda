"""Import-only stand-in for NVIDIA Isaac Gym (a closed binary that cannot exist on an AMD node).

The reference scripts start with `import isaacgym` (scripts/train.py:3-4) before torch; the MI355X stack
does not use any Isaac Gym API — physics runs in libgo1sim (HIP).  `isaacgym.torch_utils` re-exports the
quaternion helpers (restated from their definitions in go1_gym/utils/math_utils.py) because user scripts
such as scripts/play.py import them from there."""
from . import torch_utils  # noqa: F401
from . import gymapi, gymutil, gymtorch  # noqa: F401
