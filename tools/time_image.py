#!/usr/bin/env python3
"""Full-image inference (test.py -> MoCoFlowTrainer.render, trainer_moco_flow.py:226-268) through the device-side
chunk driver: H x W rays generated on the GPU (mf_make_rays), hierarchical 64 + 128 sampling, test_time=True,
canonical NeRF only and the full MoCo path (bw NoF -> NeRF), fp32 and bf16.  Usage: time_image.py [H] [W]"""
import functools, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import moco_flow_amd as M
from moco_flow_amd import camera, image, rendering, synth
rendering.STRICT_RNG = False
H = int(sys.argv[1]) if len(sys.argv) > 1 else 540
W = int(sys.argv[2]) if len(sys.argv) > 2 else 540
dev = torch.device("cuda")


def load(m, sd):
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); return m.to(dev)


c2w = np.array([[1, 0, 0, 0.0], [0, 1, 0, 0.0], [0, 0, 1, 4.0]], dtype=np.float64)
rays = camera.make_rays(H, W, 1.2 * W, (W / 2, H / 2), c2w, 2.0, 6.0, -0.25)
rays = torch.cat([rays, torch.full((H * W, 1), 0.5, device=dev)], 1)
bg = torch.ones(H * W, 3, device=dev)
msk = np.ones(H * W, dtype=bool); msk[::7] = False            # a foreground mask that drops 1/7 of the pixels
cases = {}
nerfs = [load(M.NeRF(8, 256, 63, [4], "dir", 27), synth.nerf_state(0, regime="dense", tag=t)) for t in ("coarse", "fine")]
cases["canonical NeRF"] = functools.partial(M.render_rays, nerf_embeddings=[M.Embedding(3, 10), None, M.Embedding(3, 4)], nerf_models=nerfs,
                                            N_samples=64, N_importance=128, perturb=0, noise_std=0, test_time=True)
nerfi = [load(M.NeRF(8, 256, 63, [4], "ind", 5), synth.nerf_state(0, extra_feat_type="ind", extra_feat_dim=5, regime="dense", tag=t)) for t in ("coarse", "fine")]
nofs = [load(M.NoF(4, 128, 33, [2], "ind", 33, True), synth.nof_state(0, use_quat=True, tag="bw", head_scale=0.25))]
cases["MoCo (bw NoF -> NeRF)"] = functools.partial(M.render_rays, nerf_embeddings=[M.Embedding(3, 10), M.Embedding(1, 2), None], nerf_models=nerfi,
                                                   nof_embeddings=[M.Embedding(3, 5), M.Embedding(1, 16)], nof_models=nofs,
                                                   N_samples=64, N_importance=128, perturb=0, noise_std=0, test_time=True)
n_valid = int(msk.sum())
print(f"{H} x {W} image, {n_valid} valid rays x (64 + 192) samples = {n_valid * 256 / 1e6:.1f} M network evaluations per frame")
for prec in ("f32", "bf16"):
    rendering.set_precision(prec)
    for name, fn in cases.items():
        render = lambda r, b: fn(r, b)
        with torch.no_grad():
            image.render_image(rays, bg, render, 65536, msk); torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(3):
                out = image.render_image(rays, bg, render, 65536, msk)
            torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / 3 * 1e3
        print(f"  {prec:5s} {name:24s}: {ms:8.1f} ms / frame  ({n_valid * 256 / (ms * 1e-3):.3e} ray-samples/s)")
rendering.set_precision("f32")
