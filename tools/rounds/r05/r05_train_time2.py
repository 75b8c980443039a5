import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
import moco_flow_amd as M
from moco_flow_amd import rendering, synth
rendering.STRICT_RNG = False
dev = torch.device("cuda:0")
crit = M.get_loss(dict(type="MSE"))
nerfs, nofs, rays, bg, gt, embs, kw = bench.joint_stage_setup(M, synth, torch, dev, 1024)
def joint():
    for m in nerfs + nofs:
        m.zero_grad(set_to_none=True)
    res = M.render_rays(rays, bg, embs, nerfs, **kw)
    loss = crit(res, gt)
    for k in ("nof_local_disp_coarse", "nof_global_disp_coarse", "nof_local_disp_fine", "nof_global_disp_fine"):
        loss = loss + 0.1 * res[k].mean()
    loss.backward()
for prec in ("f32", "bf16x3", "f32", "bf16x3"):
    rendering.set_train_forward_precision(prec)
    ts = []
    for _ in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter(); joint(); torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t0) * 1e3, 2))
    print(prec, ts, "reserved GB", round(torch.cuda.memory_reserved() / 1e9, 2), flush=True)
