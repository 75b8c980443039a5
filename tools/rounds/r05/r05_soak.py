"""Soak: N joint-stage training iterations (Adam step, default garbage collector) -- iteration times and the allocator's reserve at the
start and at the end (round 5: before the reference-cycle fix the reserve grew by ~9 GB per step between collections)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, bench
import moco_flow_amd as M
from moco_flow_amd import rendering, synth
rendering.STRICT_RNG = False
dev = torch.device("cuda:0")
n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 400
crit = M.get_loss(dict(type="MSE"))
nerfs, nofs, rays, bg, gt, embs, kw = bench.joint_stage_setup(M, synth, torch, dev, 1024)
opt = torch.optim.Adam([p for m in nerfs + nofs for p in m.parameters()], lr=1e-6)
ts, res0 = [], None
for i in range(n_it):
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    res = M.render_rays(rays, bg, embs, nerfs, **kw)
    loss = crit(res, gt)
    for k in ("nof_local_disp_coarse", "nof_global_disp_coarse", "nof_local_disp_fine", "nof_global_disp_fine"):
        loss = loss + 0.1 * res[k].mean()
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
    if os.environ.get("MF_SOAK_DEL") == "1":      # drop the result dict (and with it the pass and its dumps) before the next forward
        lv = float(loss)
        del res, loss
    if i == 5:
        res0 = torch.cuda.memory_reserved()
s = sorted(ts[5:])
print("iterations above 2 x the median:", [(i, round(t, 1)) for i, t in enumerate(ts) if i >= 5 and t > 2 * s[len(s) // 2]])
print(f"{n_it} iterations: median {s[len(s)//2]:.2f} ms, p99 {s[int(len(s)*0.99)]:.2f}, max {s[-1]:.2f}; reserved {res0/1e9:.2f} GB after 5, {torch.cuda.memory_reserved()/1e9:.2f} GB at the end; "
      f"peak allocated {torch.cuda.max_memory_allocated()/1e9:.2f} GB; loss {(lv if os.environ.get('MF_SOAK_DEL') == '1' else float(loss)):.5f}")
