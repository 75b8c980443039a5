import os, sys, time, warnings
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, bench
import moco_flow_amd as M
from moco_flow_amd import rendering, synth
rendering.STRICT_RNG = False
dev = torch.device("cuda:0")
crit = M.get_loss(dict(type="MSE"))
nerfs, nofs, rays, bg, gt, embs, kw = bench.joint_stage_setup(M, synth, torch, dev, 1024)
opt = torch.optim.Adam([p for m in nerfs + nofs for p in m.parameters()], lr=1e-6)
def it():
    opt.zero_grad(set_to_none=True)
    res = M.render_rays(rays, bg, embs, nerfs, **kw)
    loss = crit(res, gt)
    for k in ("nof_local_disp_coarse", "nof_global_disp_coarse", "nof_local_disp_fine", "nof_global_disp_fine"):
        loss = loss + 0.1 * res[k].mean()
    loss.backward()
    opt.step()
for _ in range(3): it()
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    it()
torch.cuda.set_sync_debug_mode("default")
print("synchronizing calls in one iteration:", len(w))
import collections
c = collections.Counter((str(x.filename).split("/")[-1], x.lineno) for x in w)
for k, v in c.most_common(20): print("  ", k, v)
# python time: profile one iteration's host side
import cProfile, pstats, io
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); it(); pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:3500])
