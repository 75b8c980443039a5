"""Debug aid: golden cases through each precision mode, PSNR vs the golden outputs."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from cases import RENDER_CASES
from helpers import build_case, load_golden
import moco_flow_amd as M
from moco_flow_amd import rendering

def psnr(a, b):
    mse = float(((a.double().cpu() - b.double()) ** 2).mean())
    return -10 * np.log10(mse) if mse > 0 else 200.0

names = sys.argv[1:] or ["r_nerf_dir_dense", "r_nerf_ind_dense", "r_nerf_none_dense", "r_moco_bw_only", "r_moco_local", "r_moco_global"]
for name in names:
    c = dict(RENDER_CASES[name]); g = load_golden(name); seed = int(g["meta_seed"])
    embs, nerfs, kw = build_case(M, c, seed, device="cuda")
    rays = torch.from_numpy(g["in_rays"]).cuda(); bg = torch.from_numpy(g["in_background"]).cuda() if c.get("bg", True) else None
    for prec in ("f32", "bf16", "bf16x3"):
        rendering.set_precision(prec)
        with torch.no_grad():
            res = M.render_rays(rays, bg, embs, nerfs, **kw)
        rendering.set_precision("f32")
        print(name, prec, {k: round(psnr(res[k], torch.from_numpy(g["out_" + k])), 1) for k in ("rgb_coarse", "depth_coarse", "opacity_coarse")}, flush=True)
