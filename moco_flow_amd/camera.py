"""Device-side ray generation: Camera.make_rays of
/root/reference/utils/camera.py:134-148 (gen_ray_directions :29-50, gen_rays :52-81) as one HIP
launch writing the (H*W, 9) ray tensor straight into GPU memory -- no host tensor, no H2D copy per
chunk (trainer_moco_flow.py:201-202)."""
import ctypes as C

import numpy as np
import torch

from . import _lib as L


def make_rays(H, W, focal, center, c2w, near, far, idx, device="cuda"):
    """rays (H*W, 9) = [o, unit d, near, far, idx] on ``device``; ``focal`` = K[0][0] (the reference uses
    focal[0] for both axes, camera.py:47), ``center`` = (K[0][2], K[1][2]), ``c2w`` (3|4, 4) array or None."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("moco_flow_amd.camera.make_rays: this is the MI355X (HIP) path; no CPU implementation")
    out = torch.empty((H * W, 9), device=dev, dtype=torch.float32)
    mat = None
    if c2w is not None:
        m = np.asarray(c2w, dtype=np.float64)[:3, :4].astype(np.float32)      # camera.py:141 .float()
        mat = (C.c_float * 12)(*m.reshape(-1).tolist())
    with torch.cuda.device(dev):
        L.check(L.lib().mf_make_rays(H, W, float(np.float32(focal)), float(np.float32(center[0])),
                                     float(np.float32(center[1])), mat, float(np.float32(near)),
                                     float(np.float32(far)), float(np.float32(idx)), out.data_ptr(),
                                     L.current_stream(dev)), "mf_make_rays")
    return out


def near_far_from_aabb(aabb_verts, c2w):
    """camera.py:138-139: min / max distance from the camera origin to the AABB corners (host scalars)."""
    d = np.sqrt(np.sum((np.asarray(aabb_verts) - np.asarray(c2w)[:3, 3]) ** 2, axis=-1))
    return float(min(d)), float(max(d))


def project_aabb(aabb_verts, c2w, K):
    """calculate_2d_projections (camera.py:83-103): the 8 AABB corners in integer pixel coordinates (x = column,
    y = row), host side (8 points), same numpy operations in the same order as the reference."""
    pts = np.asarray(aabb_verts).transpose()
    homo = np.vstack([pts, np.ones((1, pts.shape[1]), dtype=np.float32)])
    cam = np.linalg.inv(np.asarray(c2w)) @ homo
    cam = cam[:3, :] / cam[3, :]
    cam[1:, :] *= -1
    pix = np.asarray(K) @ cam[:3, :]
    pix = (pix[:2, :] / pix[2, :]).transpose()
    return np.array(pix, dtype=np.int32)


def valid_rays_mask(aabb_verts, c2w, K, size, device="cuda"):
    """Camera.get_valid_rays_mask (camera.py:119-132): bool (H*W,) on ``device`` -- the filled convex hull of the
    projected AABB, written by mf_valid_rays_mask (no H x W host image, no cv2).  ``size`` = (H, W)."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("moco_flow_amd.camera.valid_rays_mask: this is the MI355X (HIP) path; no CPU implementation")
    H, W = int(size[0]), int(size[1])
    pix = np.ascontiguousarray(project_aabb(aabb_verts, c2w, K).reshape(-1, 2), dtype=np.int32)
    out = torch.empty(H * W, device=dev, dtype=torch.uint8)
    arr = (C.c_int32 * pix.size)(*pix.reshape(-1).tolist())
    with torch.cuda.device(dev):
        L.check(L.lib().mf_valid_rays_mask(H, W, arr, pix.shape[0], out.data_ptr(), L.current_stream(dev)),
                "mf_valid_rays_mask")
    return out.bool()
