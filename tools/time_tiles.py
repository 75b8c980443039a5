#!/usr/bin/env python3
"""Kernel time of a bench configuration against the number of rays (tiles per workgroup): separates the per-launch fixed
cost (prologue: resident weights / per-ray tables -> LDS, first panels; tail) from the per-tile cost.
usage: time_tiles.py [CONFIG] [rays ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import bench
import moco_flow_amd as M
from moco_flow_amd import rendering, synth

name = sys.argv[1] if len(sys.argv) > 1 else "C3g"
sizes = [int(v) for v in sys.argv[2:]] or [1024, 2048, 3072, 4096, 6144, 8192, 16384]
cfg = dict(bench.CONFIGS[name])
dev = torch.device("cuda:0")
rendering.STRICT_RNG = False
rendering.set_precision(cfg["precision"])
models = bench.build_models(M, synth, dev, cfg)
kw = bench.render_kwargs(cfg, models)
pts = []
for n in sizes:
    cfg["rays"] = n
    r, b = synth.rays(0, n, chained=(cfg["nof"] == "global"))
    rays, bg = torch.from_numpy(r).to(dev), torch.from_numpy(b).to(dev)
    ms, samples, how = bench.kernel_probe(M, rendering, torch, cfg, models, rays, bg, kw)
    pts.append((samples, ms))
    print(f"{name} rays {n:6d}: launch of {samples:8d} samples  {ms*1e3:8.1f} us   {ms*1e6/samples:6.3f} ns/sample")
x, y = np.array([p[0] for p in pts], float), np.array([p[1] for p in pts]) * 1e3
a, c = np.polyfit(x, y, 1)
print(f"fit: {c:.1f} us fixed + {a*1e3:.3f} ns/sample")
