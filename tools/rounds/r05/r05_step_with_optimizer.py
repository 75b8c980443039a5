"""The joint-stage training iteration as a trainer runs it: render_rays + loss + backward + Adam step (every network re-packs its
weight streams at the next forward) against the step without the optimizer (what bench.py's train_ms times)."""
import os, sys, time, gc
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, bench
import moco_flow_amd as M
from moco_flow_amd import rendering, synth
rendering.STRICT_RNG = False
dev = torch.device("cuda:0")
crit = M.get_loss(dict(type="MSE"))
nerfs, nofs, rays, bg, gt, embs, kw = bench.joint_stage_setup(M, synth, torch, dev, 1024)
params = [p for m in nerfs + nofs for p in m.parameters()]
opt = torch.optim.Adam(params, lr=1e-6)
def it(with_opt):
    opt.zero_grad(set_to_none=True)
    res = M.render_rays(rays, bg, embs, nerfs, **kw)
    loss = crit(res, gt)
    for k in ("nof_local_disp_coarse", "nof_global_disp_coarse", "nof_local_disp_fine", "nof_global_disp_fine"):
        loss = loss + 0.1 * res[k].mean()
    loss.backward()
    if with_opt:
        opt.step()
for with_opt in (False, True, False, True):
    it(with_opt); it(with_opt); torch.cuda.synchronize(); gc.collect(); gc.disable()
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); it(with_opt); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    gc.enable()
    print("with Adam step + re-pack" if with_opt else "without optimizer      ", "median", round(sorted(ts)[4], 2), "ms", [round(t, 1) for t in ts], flush=True)
