"""TEST INFRASTRUCTURE ONLY. Restatement of the two kornia==0.6.5 functions that
``NoF(use_quat=True)`` calls (/root/reference/models/nof.py:4,78-79; version pin
/root/reference/docker/requirements.txt:16).

PARITY UNPINNED: kornia is a third-party dependency that is absent from
/root/reference and from this image (no network), and the reference holds no
test vector for it. What follows restates the published algorithm of
``kornia.geometry.conversions`` 0.6.5 with its default coefficient order
(``QuaternionCoeffOrder.XYZW``):

  quaternion_log_to_exp(v):   n = clamp(||v||_2, min=eps=1e-8)
                              q = (v * sin(n) / n , cos(n))        # x,y,z,w
  quaternion_to_rotation_matrix(q):
                              q <- q / max(||q||_2, 1e-12)         # F.normalize
                              tx=2x ty=2y tz=2z ; twx=tx*w ... tzz=tz*z
                              R = [[1-(tyy+tzz), txy-twz,     txz+twy],
                                   [txy+twz,     1-(txx+tzz), tyz-twx],
                                   [txz-twy,     tyz+twx,     1-(txx+tyy)]]

Both functions share one coefficient order, so the composition is independent of
XYZW-vs-WXYZ. Goldens produced through this file are labelled "kornia restated".
"""
import sys
import types

import torch


def quaternion_log_to_exp(quaternion: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    norm_q = torch.norm(quaternion, p=2, dim=-1, keepdim=True).clamp(min=eps)
    quaternion_vector = quaternion * torch.sin(norm_q) / norm_q
    quaternion_scalar = torch.cos(norm_q)
    return torch.cat([quaternion_vector, quaternion_scalar], dim=-1)


def quaternion_to_rotation_matrix(quaternion: torch.Tensor) -> torch.Tensor:
    q = torch.nn.functional.normalize(quaternion, p=2.0, dim=-1, eps=1e-12)
    x, y, z, w = torch.chunk(q, chunks=4, dim=-1)
    tx, ty, tz = 2.0 * x, 2.0 * y, 2.0 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    one = torch.tensor(1.0)
    matrix = torch.stack((
        one - (tyy + tzz), txy - twz, txz + twy,
        txy + twz, one - (txx + tzz), tyz - twx,
        txz - twy, tyz + twx, one - (txx + tyy)), dim=-1).view(-1, 3, 3)
    if len(quaternion.shape) == 1:
        matrix = torch.squeeze(matrix, dim=0)
    return matrix


def install_stub() -> None:
    """Register stub ``kornia`` modules so that /root/reference/models imports
    (used only by tests/golden/gen_golden.py, in the build container)."""
    if "kornia" in sys.modules:
        return
    k = types.ModuleType("kornia")
    g = types.ModuleType("kornia.geometry")
    c = types.ModuleType("kornia.geometry.conversions")
    c.quaternion_log_to_exp = quaternion_log_to_exp
    c.quaternion_to_rotation_matrix = quaternion_to_rotation_matrix
    g.conversions = c
    k.geometry = g
    sys.modules["kornia"] = k
    sys.modules["kornia.geometry"] = g
    sys.modules["kornia.geometry.conversions"] = c
