"""Which reference cycle keeps a training step's dump tensors alive until the cyclic collector runs (tools/rounds/r05/r05_gc_check.py showed
8.9 GB per joint step)?  One step with the collector off, then DEBUG_SAVEALL: the garbage by type and who refers to the big tensors."""
import os, sys, gc, collections
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, bench
import moco_flow_amd as M
from moco_flow_amd import rendering, synth
rendering.STRICT_RNG = False
dev = torch.device("cuda:0")
crit = M.get_loss(dict(type="MSE"))
nerfs, nofs, rays, bg, gt, embs, kw = bench.joint_stage_setup(M, synth, torch, dev, 256)
def joint():
    for m in nerfs + nofs:
        m.zero_grad(set_to_none=True)
    res = M.render_rays(rays, bg, embs, nerfs, **kw)
    loss = crit(res, gt)
    for k in ("nof_local_disp_coarse", "nof_global_disp_coarse", "nof_local_disp_fine", "nof_global_disp_fine"):
        loss = loss + 0.1 * res[k].mean()
    loss.backward()
joint(); joint(); gc.collect()
gc.disable()
joint()
torch.cuda.synchronize()
gc.set_debug(gc.DEBUG_SAVEALL)
n = gc.collect()
print("garbage objects:", n, len(gc.garbage))
print(collections.Counter(type(o).__name__ for o in gc.garbage).most_common(25))
big = [o for o in gc.garbage if isinstance(o, torch.Tensor) and o.is_cuda and o.numel() * o.element_size() > 50e6]
print("big tensors in garbage:", [(tuple(t.shape), round(t.numel() * t.element_size() / 1e6)) for t in big][:12])
ids = {id(o) for o in gc.garbage}
for t in big[:3]:
    print("== referrers of", tuple(t.shape))
    for r in gc.get_referrers(t):
        if id(r) in ids:
            desc = type(r).__name__
            if isinstance(r, dict):
                desc += " keys=" + str(list(r.keys())[:12])
            elif isinstance(r, (list, tuple)):
                desc += f" len={len(r)} of " + str([type(x).__name__ for x in r[:6]])
            print("   ", desc)
            for r2 in gc.get_referrers(r):
                if id(r2) in ids and r2 is not t:
                    d2 = type(r2).__name__
                    if isinstance(r2, dict):
                        d2 += " keys=" + str(list(r2.keys())[:12])
                    print("        <-", d2)
# functions / cells in the garbage: closures that close over themselves
for o in gc.garbage:
    if type(o).__name__ == "function":
        print("function in garbage:", o.__qualname__, "at", o.__code__.co_filename.split("/")[-1], o.__code__.co_firstlineno)
