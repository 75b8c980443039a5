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
