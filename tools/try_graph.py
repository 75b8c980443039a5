"""Experiment: a render_rays pass (loss fast path: no host sync) captured in a HIP graph through torch.cuda.CUDAGraph
and replayed -- step time vs the eager call.  usage: try_graph.py [C3|C3g|C5|C2]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import bench
import moco_flow_amd as M
from moco_flow_amd import rendering, synth

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
cfg = bench.CONFIGS[name]
dev = torch.device("cuda:0")
rendering.set_precision(cfg["precision"])
rendering.STRICT_RNG = False
models = bench.build_models(M, synth, dev, cfg)
kw = bench.render_kwargs(cfg, models)
r, b = synth.rays(0, cfg["rays"], chained=(cfg["nof"] == "global"))
rays, bg = torch.from_numpy(r).to(dev), torch.from_numpy(b).to(dev)
gt = torch.rand(cfg["rays"], 3, device=dev)
lt = gt if cfg["nof"] else None


def step():
    with torch.no_grad():
        return M.render_rays(rays, bg, models["embs"], models["nerfs"], _loss_target=lt, **kw)


def timeit(f, n=300):
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


print(f"{name}: eager (pipelined, no sync) {timeit(step):.4f} ms/step")
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = step()
ref = {k: v.clone() for k, v in step().items() if torch.is_tensor(v)}
g.replay()
torch.cuda.synchronize()
for k, v in ref.items():
    assert torch.equal(out[k], v), k
print(f"{name}: graph replay {timeit(g.replay):.4f} ms/step  (outputs identical to the eager call: {sorted(ref)})")
