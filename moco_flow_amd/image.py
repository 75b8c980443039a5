"""Full-image chunk driver on the device: the ``render`` method of the reference's
trainers (trainer/trainer_moco_flow.py:226-268, trainer/trainer_nerf.py:100-140) without its host
round trips -- the valid-ray selection, the per-chunk loop and the foreground scatter-back all stay in
GPU memory (the reference indexes with numpy masks and pulls ``opacity`` to the host, :255)."""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, Optional

import torch

from . import _lib as L


def compose_image(rays_msk: Optional[torch.Tensor], opacity, rgb, depth, background):
    """mf_image_compose: (img (B,3), depth (B,)) from the rendered rays' colour / depth / opacity."""
    L.require_gpu(background, "image.compose_image")
    dev = background.device
    B = background.shape[0]
    img = torch.empty((B, 3), device=dev, dtype=torch.float32)
    dep = torch.empty((B,), device=dev, dtype=torch.float32)
    msk8 = rank = None
    if rays_msk is not None:
        m = torch.as_tensor(rays_msk, device=dev).reshape(-1).bool()
        msk8 = m.to(torch.uint8).contiguous()
        rank = (torch.cumsum(m.to(torch.int64), 0) - 1).contiguous()
    f = lambda t: t.detach().contiguous().float()
    opacity, rgb, depth, bgc = f(opacity), f(rgb), f(depth), f(background)
    with torch.cuda.device(dev):
        L.check(L.lib().mf_image_compose(msk8.data_ptr() if msk8 is not None else None,
                                         rank.data_ptr() if rank is not None else None, B, opacity.data_ptr(),
                                         rgb.data_ptr(), depth.data_ptr(), bgc.data_ptr(), img.data_ptr(),
                                         dep.data_ptr(), L.current_stream(dev)), "mf_image_compose")
    return img, dep


def render_image(rays, background, render: Callable[..., Dict[str, torch.Tensor]], N_rand: int,
                 rays_msk=None) -> Dict[str, torch.Tensor]:
    """``render(rays_chunk, background_chunk) -> result dict`` (e.g. a ``functools.partial`` of
    ``render_rays``) applied to the valid rays in chunks of ``N_rand`` and, when a mask is given,
    scattered back into full-image ``rgb_*`` / ``depth_*`` exactly as trainer_moco_flow.py:249-266 does:
    every other key stays per rendered ray.  ``rays`` / ``background`` / ``rays_msk`` may live on the host
    (they are moved once, not per chunk)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if not (torch.is_tensor(rays) and rays.is_cuda) else rays.device
    rays = torch.as_tensor(rays).to(dev)
    background = torch.as_tensor(background).to(dev)
    sel_rays, sel_bg, m = rays, background, None
    if rays_msk is not None:
        m = torch.as_tensor(rays_msk).to(dev).reshape(-1).bool()
        sel_rays, sel_bg = rays[m], background[m]
    chunks = []
    for i in range(0, sel_rays.shape[0], N_rand):
        chunks.append(render(sel_rays[i:i + N_rand], sel_bg[i:i + N_rand]))
    if not chunks:                                   # no valid ray at all: the reference's empty-chunk result
        chunks.append(render(sel_rays[:0], sel_bg[:0]))
    results = {k: torch.cat([c[k] for c in chunks], 0) for k in chunks[0]}
    if rays_msk is not None:
        typ = "fine" if "rgb_fine" in results else "coarse"
        img, dep = compose_image(m, results[f"opacity_{typ}"], results[f"rgb_{typ}"], results[f"depth_{typ}"], background)
        results[f"rgb_{typ}"] = img
        results[f"depth_{typ}"] = dep
    return results
