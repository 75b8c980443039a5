"""Consistency sweep over ragged launch shapes: N in {1 .. 5000} x S in {40, 64, 128, 192}, fp32 NeRF / fp32 MoCo chain /
bf16 MoCo global chain / bf16x3 NeRF, local and global chains -- one launch must equal the concatenation of two launches over a split of the rays BIT FOR BIT (ray
groups, tiles per group and the composite phase are launch-shape dependent; per-ray results must not be)."""
import sys, os, itertools
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch, numpy as np
import bench
import moco_flow_amd as M
from moco_flow_amd import rendering, synth
rendering.STRICT_RNG = False
dev = torch.device("cuda:0")
bad = 0
for prec, cfgname in (("f32", "C2"), ("f32", "C3f"), ("bf16", "C3g"), ("bf16x3", "C2x"), ("bf16x3", "C3x"), ("bf16x3", "C5x")):
    cfg = dict(bench.CONFIGS[cfgname]); cfg["precision"] = prec
    rendering.set_precision(prec)
    models = bench.build_models(M, synth, dev, cfg)
    for N, S in itertools.product((1, 3, 100, 257, 1000, 4097, 5000), (64, 40, 128, 192)):
        r, b = synth.rays(0, N, chained=(cfg["nof"] == "global"))
        rays, bg = torch.from_numpy(r).to(dev), torch.from_numpy(b).to(dev)
        kw = bench.render_kwargs(cfg, models); kw["N_samples"] = S
        with torch.no_grad():
            full = M.render_rays(rays, bg, models["embs"], models["nerfs"], test_time=True, **kw)
            h = max(1, N // 3)
            a = M.render_rays(rays[:h], bg[:h], models["embs"], models["nerfs"], test_time=True, **kw)
            c = M.render_rays(rays[h:], bg[h:], models["embs"], models["nerfs"], test_time=True, **kw) if N > h else None
        for k in full:
            cat = torch.cat([a[k], c[k]]) if c is not None else a[k]
            if not torch.equal(full[k], cat):
                bad += 1
                print("MISMATCH", cfgname, prec, N, S, k, float((full[k] - cat).abs().max()))
print("ragged sweep done, mismatches:", bad)
