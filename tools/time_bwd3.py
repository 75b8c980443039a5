#!/usr/bin/env python3
"""Time mf_nerf_backward3 / mf_nerf_backward_x alone on a stage-1 sized dump (P = 5120 x 256 samples)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import moco_flow_amd as M
from moco_flow_amd import autograd as A

P = int(sys.argv[1]) if len(sys.argv) > 1 else 5120 * 256
dev = torch.device("cuda")
nerf = M.NeRF(8, 256, 63, [4], "dir", 27).cuda()
stride = 9 * 256 + 128
acts = torch.randn(P, stride, device=dev).relu_()
rgbsig = torch.rand(P, 4, device=dev)
g_out = torch.randn(P, 4, device=dev)
mask = torch.randint(-2**31, 2**31 - 1, (P, 80), device=dev, dtype=torch.int32)
for prec, with_mask in (("f32", False), ("bf16x3", False), ("bf16x3", True)):
    A.set_dx_precision(prec)
    if with_mask:
        acts._mf_mask = mask
    elif hasattr(acts, "_mf_mask"):
        del acts._mf_mask
    A.nerf_backward_hip(nerf, g_out, acts, rgbsig); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        A.nerf_backward_hip(nerf, g_out, acts, rgbsig)
    torch.cuda.synchronize()
    print(f"dX chain {prec}{' + bit masks' if with_mask else ''}: {(time.perf_counter() - t) / 5 * 1e3:.3f} ms  (P = {P})")
