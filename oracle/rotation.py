"""Oracle: rotation conversions on the hot path (test infrastructure, see oracle/__init__.py).

Restates mogen/models/utils/rotation_conversions.py (an old PyTorch3D fork: quaternion
route with `_sqrt_positive_part`/`_copysign`).
"""
import torch
import torch.nn.functional as F


def axis_angle_to_quaternion(aa):
    """reference: rotation_conversions.py:450-479"""
    angles = torch.norm(aa, p=2, dim=-1, keepdim=True)
    half = 0.5 * angles
    small = angles.abs() < 1e-6
    s = torch.where(small, 0.5 - (angles * angles) / 48,
                    torch.sin(half) / torch.where(small, torch.ones_like(angles), angles))
    return torch.cat([torch.cos(half), aa * s], dim=-1)


def quaternion_to_matrix(q):
    """reference: rotation_conversions.py:36-64"""
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((
        1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
        two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
        two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def axis_angle_to_matrix(aa):
    """reference: rotation_conversions.py:416-430"""
    return quaternion_to_matrix(axis_angle_to_quaternion(aa))


def matrix_to_rotation_6d(m):
    """reference: rotation_conversions.py:535-550"""
    return m[..., :2, :].clone().reshape(*m.size()[:-2], 6)


def rotation_6d_to_matrix(d6):
    """reference: rotation_conversions.py:511-532"""
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = F.normalize(a1, dim=-1)
    b2 = a2 - (b1 * a2).sum(-1, keepdim=True) * b1
    b2 = F.normalize(b2, dim=-1)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-2)


def _sqrt_positive_part(x):
    """reference: rotation_conversions.py:84-92"""
    return torch.where(x > 0, torch.sqrt(torch.clamp(x, min=0)), torch.zeros_like(x))


def _copysign(a, b):
    """reference: rotation_conversions.py:67-81"""
    return torch.where((a < 0) != (b < 0), -a, a)


def matrix_to_quaternion(m):
    """reference: rotation_conversions.py:95-118"""
    m00, m11, m22 = m[..., 0, 0], m[..., 1, 1], m[..., 2, 2]
    o0 = 0.5 * _sqrt_positive_part(1 + m00 + m11 + m22)
    x = 0.5 * _sqrt_positive_part(1 + m00 - m11 - m22)
    y = 0.5 * _sqrt_positive_part(1 - m00 + m11 - m22)
    z = 0.5 * _sqrt_positive_part(1 - m00 - m11 + m22)
    o1 = _copysign(x, m[..., 2, 1] - m[..., 1, 2])
    o2 = _copysign(y, m[..., 0, 2] - m[..., 2, 0])
    o3 = _copysign(z, m[..., 1, 0] - m[..., 0, 1])
    return torch.stack((o0, o1, o2, o3), -1)


def quaternion_to_axis_angle(q):
    """reference: rotation_conversions.py:482-508"""
    norms = torch.norm(q[..., 1:], p=2, dim=-1, keepdim=True)
    half = torch.atan2(norms, q[..., :1])
    angles = 2 * half
    small = angles.abs() < 1e-6
    s = torch.where(small, 0.5 - (angles * angles) / 48,
                    torch.sin(half) / torch.where(small, torch.ones_like(angles), angles))
    return q[..., 1:] / s


def matrix_to_axis_angle(m):
    """reference: rotation_conversions.py:433-447"""
    return quaternion_to_axis_angle(matrix_to_quaternion(m))


def aa_to_6d(x, joints):
    """[B,N,J*3] -> [B,N,J*6] (callers: diffusion_transformer.py:193-224)."""
    b, n, _ = x.shape
    return matrix_to_rotation_6d(axis_angle_to_matrix(x.reshape(b, n, joints, 3))).reshape(b, n, joints * 6)


def sixd_to_aa(x, joints):
    """[B,N,J*6] -> [B,N,J*3] (callers: diffusion_transformer.py:296-328)."""
    b, n, _ = x.shape
    return matrix_to_axis_angle(rotation_6d_to_matrix(x.reshape(b, n, joints, 6))).reshape(b, n, joints * 3)
