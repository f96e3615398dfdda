import numpy as np  # noqa: F401
import torch  # noqa: F401

from go1_gym.utils.math_utils import (get_axis_params, normalize, quat_apply, quat_conjugate,  # noqa: F401
                                      quat_from_angle_axis, quat_mul, quat_rotate, quat_rotate_inverse, to_torch,
                                      torch_rand_float)
