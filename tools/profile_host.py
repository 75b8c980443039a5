import sys, os, time, cProfile, pstats
sys.path.insert(0, os.getcwd())
import torch, bench
import moco_flow_amd as M
from moco_flow_amd import synth, rendering
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "C3"]
dev = torch.device("cuda:0")
models = bench.build_models(M, synth, dev, cfg)
kw = bench.render_kwargs(cfg, models)
r, b = synth.rays(0, cfg["rays"], chained=(cfg["nof"] == "global"))
rays, bg = torch.from_numpy(r).to(dev), torch.from_numpy(b).to(dev)
rendering.set_precision(cfg["precision"])
rendering.STRICT_RNG = False
MEANS = os.environ.get("MF_MEANS", "1") == "1"        # the trainer's use of the consensus vectors (as bench.py's step)
def step():
    with torch.no_grad():
        out = M.render_rays(rays, bg, models["embs"], models["nerfs"], **kw)
        if MEANS:
            cons = [torch.mean(v) for k, v in out.items() if k.startswith("nof_")]
        return out
for _ in range(20): step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(200): step()
th = time.perf_counter()
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t) / 200 * 1e3, " host enqueue ms/step", (th - t) / 200 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
