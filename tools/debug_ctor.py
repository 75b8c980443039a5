"""debug: NeRF() default-constructor gradients under the backward arithmetics (GPU)"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import moco_flow_amd as M
from moco_flow_amd import synth, autograd as A
from oracle import cpu_ref as R
from helpers import relerr

for cx, nf in ((33, 5), (63, 10)):
    n, S = 96, 64
    rays_np, bg_np = synth.rays(3, n)
    rays, bg = torch.from_numpy(rays_np), torch.from_numpy(bg_np)
    nerf = M.NeRF(8, 256, cx, [4], "none", 0)
    nerf.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_state(3, in_channels_xyz=cx, extra_feat_type="none", extra_feat_dim=0, regime="dense", tag="ctor").items()})
    nerf = nerf.cuda()
    sd = {k: v.detach().cpu().clone() for k, v in nerf.state_dict().items()}
    embs = [M.Embedding(3, nf), None, None]
    gt = torch.rand(n, 3, generator=torch.Generator().manual_seed(0))
    kw = dict(N_samples=S, noise_std=0)
    o = R.NeRF(8, 256, cx, [4], "none", 0, state=sd)
    for k in o.p:
        o.p[k] = o.p[k].clone().requires_grad_(True)
    r = R.render_rays(rays, bg, [R.Embedding(3, nf), None, None], [o], **kw)
    loss = ((r["rgb_coarse"] - gt) ** 2).mean() + 0.1 * r["depth_coarse"].mean()
    names = list(o.p)
    want = dict(zip(names, torch.autograd.grad(loss, [o.p[k] for k in names])))
    for wg, dx in (("f32", "f32"), ("bf16x3", "f32"), ("f32", "bf16x3"), ("bf16x3", "bf16x3")):
        A.set_wgrad_precision(wg); A.set_dx_precision(dx)
        nerf.zero_grad(set_to_none=True)
        res = M.render_rays(rays.cuda(), bg.cuda(), embs, [nerf], **kw)
        (((res["rgb_coarse"] - gt.cuda()) ** 2).mean() + 0.1 * res["depth_coarse"].mean()).backward()
        errs = {k: relerr(p.grad, want[k]) for k, p in nerf.named_parameters()}
        worst = max(errs, key=errs.get)
        print(f"cx={cx} wgrad={wg} dx={dx}: worst {worst} {errs[worst]:.2e}; " + " ".join(f"{errs[k]:.1e}" for k in names if k.endswith("weight")))
