import os, sys, gc
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, bench
import moco_flow_amd as M
from moco_flow_amd import rendering, synth
rendering.STRICT_RNG = False
dev = torch.device("cuda:0")
nerfs, nofs, rays, bg, gt, embs, kw = bench.joint_stage_setup(M, synth, torch, dev, 1024)
for prec in ("f32", "bf16", "bf16x3"):
    rendering.set_precision(prec)
    def f():
        with torch.no_grad():
            res = M.render_rays(rays, bg, embs, nerfs, **kw)
            s = res["rgb_fine"].mean() + res["nof_local_disp_fine"].mean() + res["nof_global_disp_coarse"].mean()
        return float(s)
    f(); f(); gc.collect(); base = torch.cuda.memory_allocated(); gc.disable()
    for i in range(3):
        f(); torch.cuda.synchronize()
        print(prec, "inference pass", i, "allocated", round(torch.cuda.memory_allocated() / 1e6, 1), "MB (base", round(base / 1e6, 1), ")", flush=True)
    gc.enable()
