"""Quaternion/angle helpers used around the env (reference go1_gym/utils/math_utils.py:12-38 and the
`isaacgym.torch_utils` functions it relies on, restated from their mathematical definitions —
SURVEY.md App. E: quaternions are xyzw)."""
import math

import torch


def normalize(x, eps: float = 1e-9):
    return x / x.norm(p=2, dim=-1).clamp(min=eps).unsqueeze(-1)


def quat_apply(q, v):
    """R(q) v for xyzw quaternions; broadcasts over leading dims."""
    shape = v.shape
    q = q.reshape(-1, 4)
    v = v.reshape(-1, 3)
    u = q[:, :3]
    t = 2.0 * torch.cross(u, v, dim=-1)
    return (v + q[:, 3:4] * t + torch.cross(u, t, dim=-1)).view(shape)


def quat_rotate(q, v):
    return quat_apply(q, v)


def quat_conjugate(q):
    return torch.cat((-q[..., :3], q[..., 3:4]), dim=-1)


def quat_rotate_inverse(q, v):
    return quat_apply(quat_conjugate(q), v)


def quat_mul(a, b):
    ax, ay, az, aw = a.unbind(-1)
    bx, by, bz, bw = b.unbind(-1)
    return torch.stack((aw * bx + ax * bw + ay * bz - az * by,
                        aw * by - ax * bz + ay * bw + az * bx,
                        aw * bz + ax * by - ay * bx + az * bw,
                        aw * bw - ax * bx - ay * by - az * bz), dim=-1)


def quat_from_angle_axis(angle, axis):
    half = (angle * 0.5).unsqueeze(-1)
    return torch.cat((normalize(axis) * half.sin(), half.cos()), dim=-1)


def quat_apply_yaw(quat, vec):
    qy = quat.clone().view(-1, 4)
    qy[:, :2] = 0.
    return quat_apply(normalize(qy), vec)


def wrap_to_pi(angles):
    angles %= 2 * math.pi
    angles -= 2 * math.pi * (angles > math.pi)
    return angles


def torch_rand_float(lower, upper, shape, device):
    return (upper - lower) * torch.rand(*shape, device=device) + lower


def torch_rand_sqrt_float(lower, upper, shape, device):
    r = 2 * torch.rand(*shape, device=device) - 1
    r = torch.where(r < 0., -torch.sqrt(-r), torch.sqrt(r))
    return (upper - lower) * (r + 1.) / 2. + lower


def get_scale_shift(range):          # noqa: A002  (the reference's keyword name, math_utils.py:35)
    lo, hi = range[0], range[1]
    return 2. / (hi - lo), (hi + lo) / 2.


def to_torch(x, dtype=torch.float, device='cpu', requires_grad=False):
    return torch.tensor(x, dtype=dtype, device=device, requires_grad=requires_grad)


def get_axis_params(value, axis_idx, x_value=0., dtype=float, n_dims=3):
    out = [0.] * n_dims
    out[axis_idx] = value
    out[0] = out[0] if axis_idx == 0 else x_value
    return out
