#!/usr/bin/env python3
"""Run the same render_rays call repeatedly and report bitwise differences (race screen for the fused kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch
import moco_flow_amd as M
from moco_flow_amd import rendering, synth
from cases import RENDER_CASES
from helpers import build_case

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rendering.STRICT_RNG = False
rendering.set_precision(prec)
for name, n in (("r_nerf_dir_dense", 4096), ("r_moco_local", 4096), ("r_moco_global", 4096), ("r_moco_global_fine", 1024)):
    c = dict(RENDER_CASES[name])
    rays_np, bg_np = synth.rays(0, n, chained=(c.get("nof") == "global"))
    rays, bg = torch.from_numpy(rays_np).cuda(), torch.from_numpy(bg_np).cuda()
    embs, nerfs, kw = build_case(M, c, 0, device="cuda")
    with torch.no_grad():
        ref = M.render_rays(rays, bg, embs, nerfs, **kw)
        bad = {}
        for r in range(reps):
            out = M.render_rays(rays, bg, embs, nerfs, **kw)
            for k in ref:
                if ref[k].shape != out[k].shape:
                    bad.setdefault(k, []).append((r, "shape", tuple(out[k].shape)))
                elif not torch.equal(ref[k], out[k]):
                    d = (ref[k] - out[k]).abs().reshape(ref[k].shape[0], -1).amax(1)
                    idx = torch.nonzero(d > 0).view(-1)
                    bad.setdefault(k, []).append((r, int(idx.numel()), float(d.max()), idx[:12].tolist()))
    brief = {k: (len(v), str(v[0])[:80]) for k, v in bad.items()}
    print(f"{prec} {name} n={n}: " + ("deterministic" if not bad else f"DIFFERS (plane: runs that differ, first) {brief}"), flush=True)

# training: the three-product forward with dumps (opt-in), the three-product dX chain and weight gradients (defaults)
if prec == "bf16x3":
    rendering.set_precision("f32")
    for fwd in ("f32", "bf16x3"):
        rendering.set_train_forward_precision(fwd)
        for name, n in (("r_nerf_dir_dense", 4096), ("r_moco_global", 1024)):
            c = dict(RENDER_CASES[name])
            rays_np, bg_np = synth.rays(0, n, chained=(c.get("nof") == "global"))
            rays, bg = torch.from_numpy(rays_np).cuda(), torch.from_numpy(bg_np).cuda()
            embs, nerfs, kw = build_case(M, c, 0, device="cuda")
            nets = list(nerfs) + (list(kw["nof_models"]) if kw.get("nof_models") else [])
            ref, bad = None, 0
            for r in range(max(4, reps // 3)):
                for m in nets:
                    m.zero_grad(set_to_none=True)
                res = M.render_rays(rays, bg, embs, nerfs, **kw)
                (res["rgb_coarse"].square().mean() + res["depth_coarse"].mean()).backward()
                cur = [p.grad.clone() for m in nets for p in m.parameters() if p.grad is not None] + [res["rgb_coarse"].detach().clone()]
                if ref is None:
                    ref = cur
                elif not all(torch.equal(a, b) for a, b in zip(ref, cur)):
                    bad += 1
            print(f"training step (forward {fwd}, backward three-product) {name} n={n}: " + ("deterministic" if not bad else f"DIFFERS in {bad} runs"), flush=True)
        rendering.set_train_forward_precision("f32")
