#!/usr/bin/env python3
"""Render the BASELINE MoCo / NeRF workloads in the fast bf16 mode and save every output: run once per MF_BF16_BLOCKS value
(the library reads it once per process) and compare the files with --compare A B (bitwise)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    if sys.argv[1] == "--compare":
        a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
        bad = 0
        for k in a:
            same = torch.equal(a[k], b[k]) or bool(((a[k] == b[k]) | (a[k].isnan() & b[k].isnan())).all())
            if not same:
                bad += 1
                d = (a[k].double() - b[k].double()).abs().max().item()
                print(f"DIFFERENT {k}: max abs {d:.3e} of {a[k].double().abs().max().item():.3e}")
        print(f"{len(a)} tensors compared, {bad} different")
        sys.exit(1 if bad else 0)
    import bench
    import moco_flow_amd as M
    from moco_flow_amd import rendering, synth
    rendering.STRICT_RNG = False
    dev = torch.device("cuda", 0)
    out = {}
    for name, n in (("C2b", 4096), ("C3", 4096), ("C3g", 1000), ("C5", 1024)):
        cfg = bench.CONFIGS[name]
        rendering.set_precision(cfg["precision"])
        models = bench.build_models(M, synth, dev, cfg)
        rays_np, bg_np = synth.rays(0, n, chained=(cfg["nof"] == "global"))
        rays, bg = torch.from_numpy(rays_np).to(dev), torch.from_numpy(bg_np).to(dev)
        kw = bench.render_kwargs(cfg, models)
        rendering.LAZY_CONSENSUS = False
        with torch.no_grad():
            res = M.render_rays(rays, bg, models["embs"], models["nerfs"], **kw)
        for k, v in res.items():
            out[f"{name}.{k}"] = v.detach().float().cpu()
    torch.save(out, sys.argv[1])
    print("saved", len(out), "tensors to", sys.argv[1])


if __name__ == "__main__":
    main()
