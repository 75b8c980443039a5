import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch
import moco_flow_amd as M
from moco_flow_amd import rendering, synth
dev = torch.device("cuda:0")
load = lambda m, sd: (m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}), m.to(dev))[1]
nerfs = [load(M.NeRF(8, 256, 63, [4], "dir", 27), synth.nerf_state(0, regime="dense", tag=t)) for t in ("coarse", "fine")]
embs = [M.Embedding(3, 10), None, M.Embedding(3, 4)]
r, b = synth.rays(0, 5120)
rays, bg = torch.from_numpy(r).to(dev), torch.from_numpy(b).to(dev)
gt = torch.rand(5120, 3, device=dev)
crit = M.get_loss(dict(type="MSE"))
opt = torch.optim.Adam([p for m in nerfs for p in m.parameters()], lr=1e-6)
ts = []
for i in range(30):
    t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True)
    res = M.render_rays(rays, bg, embs, nerfs, N_samples=128, N_importance=128, noise_std=0, perturb=0)
    loss = crit(res, gt)
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print(f"stage 1, 30 iterations keeping `res` / `loss` until overwritten: median {sorted(ts[3:])[13]:.2f} ms, peak allocated {torch.cuda.max_memory_allocated()/1e9:.2f} GB, reserved {torch.cuda.memory_reserved()/1e9:.2f} GB")
