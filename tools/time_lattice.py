#!/usr/bin/env python3
"""Mesh-extraction lattice query (visualize_mesh, trainer_moco_flow.py:485-548 / trainer_nerf.py:215-245: N_grid^3
points through [bw NoF ->] embed -> NeRF(sigma_only) in 10 000-point chunks in the reference) as ONE fused launch
per lattice (mf_points_sigma).  Usage: time_lattice.py [N_grid]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import moco_flow_amd as M
from moco_flow_amd import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda")
load = lambda m, sd: (m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}), m.to(dev))[1]
nerf = load(M.NeRF(8, 256, 63, [4], "ind", 5), synth.nerf_state(0, extra_feat_type="ind", extra_feat_dim=5, regime="dense"))
nof = load(M.NoF(4, 128, 33, [2], "ind", 33, True), synth.nof_state(0, use_quat=True, tag="bw", head_scale=0.25))
ax = torch.linspace(-1.2, 1.2, N, device=dev)
xyz = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3).contiguous()
emb = M.Embedding(3, 10)
nof_embs = [M.Embedding(3, 5), M.Embedding(1, 16)]


def timeit(f, n=3):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3


print(f"{N}^3 = {xyz.shape[0] / 1e6:.1f} M lattice points")
for prec in ("f32", "bf16", "bf16x3"):
    ms = timeit(lambda: M.query_sigma(xyz, nerf, emb, precision=prec))
    print(f"  {prec:4s} canonical space (NeRF sigma)        : {ms:8.1f} ms  {xyz.shape[0] / (ms * 1e-3):.3e} points/s")
    ms = timeit(lambda: M.query_sigma(xyz, nerf, emb, nof, nof_embs, 0.25, precision=prec))
    print(f"  {prec:4s} observation space (bw NoF -> sigma) : {ms:8.1f} ms  {xyz.shape[0] / (ms * 1e-3):.3e} points/s")
