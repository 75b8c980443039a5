"""Host time of a joint-stage training iteration: N iterations enqueued without a device sync in between (the host may run ahead of the
GPU), enqueue time per iteration against the device-bound time per iteration of the same loop."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, bench
import moco_flow_amd as M
from moco_flow_amd import rendering, synth
rendering.STRICT_RNG = False
dev = torch.device("cuda:0")
crit = M.get_loss(dict(type="MSE"))
NR = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nerfs, nofs, rays, bg, gt, embs, kw = bench.joint_stage_setup(M, synth, torch, dev, NR)
opt = torch.optim.Adam([p for m in nerfs + nofs for p in m.parameters()], lr=1e-6)
def it():
    opt.zero_grad(set_to_none=True)
    res = M.render_rays(rays, bg, embs, nerfs, **kw)
    loss = crit(res, gt)
    for k in ("nof_local_disp_coarse", "nof_global_disp_coarse", "nof_local_disp_fine", "nof_global_disp_fine"):
        loss = loss + 0.1 * res[k].mean()
    loss.backward()
    opt.step()
for _ in range(5):
    it()
torch.cuda.synchronize()
n = 40
t0 = time.perf_counter()
for _ in range(n):
    it()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{n} iterations: enqueue {1e3 * (t1 - t0) / n:.2f} ms per iteration (host), until the device is done {1e3 * (t2 - t0) / n:.2f} ms per iteration")
