"""Fused point queries: what the trainers spell as forward_nof -> embed ->
zero-pad -> NeRF(sigma_only=True) per 10 000-point chunk (trainer_moco_flow.py:146-187, 500-526;
trainer_nerf.py:215-245) as ONE launch over all points (mf_points_sigma)."""
import ctypes as C

import torch

from . import _lib as L


def query_sigma(xyz, nerf, nerf_embedding_xyz, bw_nof=None, nof_embeddings=None, ind=None, return_canonical=False,
                precision=None):
    """xyz (B,3) observation-space points -> raw sigma (B,1) of the canonical NeRF.

    bw_nof / nof_embeddings=[xyz, ind] / ind: optional backward flow at image index ``ind`` (a python
    float in [-1,1) for all points, or a (B,) / (B,1) tensor); without them the query is made in
    canonical space (visualize_mesh with frame_idx == -1). Inference only.
    ``precision``: "f32" | "bf16" | "bf16x3" (None = the module setting of ``rendering.set_precision``); bf16 = hidden GEMMs
    on the bf16 matrix pipe as in render_rays' gradient-free passes (the mesh-extraction lattice ~8x faster); bf16x3 = the
    fp32-class three-product kernels (scalar ``ind`` or no NoF; with a per-point ``ind`` tensor the exact-fp32 kernel runs
    instead, so the setting never costs accuracy)."""
    from . import rendering
    prec = L.PRECISIONS[precision or rendering.PRECISION]
    if prec == L.MF_PREC_BF16X3 and bw_nof is not None and torch.is_tensor(ind) and ind.numel() > 1:
        prec = L.MF_PREC_F32
    L.require_gpu(xyz, "query_sigma")
    x = xyz.detach().float().contiguous()
    B = x.shape[0]
    dev = x.device
    sigma = torch.empty((B, 1), device=dev, dtype=torch.float32)
    canon = torch.empty((B, 3), device=dev, dtype=torch.float32) if (return_canonical and bw_nof is not None) else None
    nd, nb = nerf.packed(prec)
    ex = nerf_embedding_xyz.descriptor()
    fd = fb = fx = fi = None
    ind_t, ind_s = None, 0.0
    if bw_nof is not None:
        fd, fb = bw_nof.packed(prec)
        fx, fi = nof_embeddings[0].descriptor(), nof_embeddings[1].descriptor()
        if torch.is_tensor(ind):
            ind_t = ind.detach().float().reshape(-1).contiguous().to(dev)
            if ind_t.numel() == 1:
                ind_s, ind_t = float(ind_t.item()), None
            elif ind_t.numel() != B:
                raise RuntimeError(f"query_sigma: ind must have 1 or {B} elements")
        else:
            ind_s = float(ind)
    need = int(L.lib().mf_points_sigma_workspace_bytes(prec, fd, 1 if ind_t is not None else 0, B)) if fd is not None else 0
    ws = torch.empty(need, dtype=torch.uint8, device=dev) if need > 0 else None     # bf16 + NoF: per-point index bias
    with torch.cuda.device(dev):
        L.check(L.lib().mf_points_sigma_p(prec, nd, nb.data_ptr(), C.byref(ex), fd, L.ptr(fb),
                                          C.byref(fx) if fx is not None else None,
                                          C.byref(fi) if fi is not None else None, L.ptr(x), L.ptr(ind_t), ind_s, B,
                                          L.ptr(sigma), L.ptr(canon), L.ptr(ws), need, L.current_stream(dev)), "mf_points_sigma")
    return (sigma, canon) if return_canonical else sigma
