"""Does a training step leave device memory to the cyclic garbage collector?  Steps with the collector OFF: allocated bytes must return to
the base after every step (round 5: a closure cycle in the consensus bookkeeping held 8.9 GB per joint step until a collection)."""
import os, sys, gc, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import torch, bench
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
for prec in ("f32", "bf16x3"):
    rendering.set_train_forward_precision(prec)
    joint(); joint(); torch.cuda.synchronize(); gc.collect(); torch.cuda.empty_cache()
    base = torch.cuda.memory_allocated()
    gc.disable()
    for i in range(6):
        t0 = time.perf_counter(); joint(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        print(prec, "step", i, f"{dt:7.2f} ms  allocated {torch.cuda.memory_allocated()/1e9:6.2f} GB (base {base/1e9:.2f})  reserved {torch.cuda.memory_reserved()/1e9:6.2f} GB", flush=True)
    n = gc.collect(); gc.enable()
    print(prec, "gc.collect() found", n, "objects; allocated after", round(torch.cuda.memory_allocated()/1e9, 2), "GB", flush=True)

# stage 1 (NeRF only, 5120 x 384)
del nerfs, nofs, rays, bg, gt
torch.cuda.empty_cache()
load = lambda m, sd: (m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}), m.to(dev))[1]
nerfs1 = [load(M.NeRF(8, 256, 63, [4], "dir", 27), synth.nerf_state(0, regime="dense", tag=t)) for t in ("coarse", "fine")]
embs1 = [M.Embedding(3, 10), None, M.Embedding(3, 4)]
r, b = synth.rays(0, 5120)
rays1, bg1 = torch.from_numpy(r).to(dev), torch.from_numpy(b).to(dev)
gt1 = torch.rand(5120, 3, device=dev)
def stage1():
    for m in nerfs1:
        m.zero_grad(set_to_none=True)
    crit(M.render_rays(rays1, bg1, embs1, nerfs1, N_samples=128, N_importance=128, noise_std=0, perturb=0), gt1).backward()
rendering.set_train_forward_precision("f32")
stage1(); stage1(); torch.cuda.synchronize(); gc.collect()
base = torch.cuda.memory_allocated()
gc.disable()
for i in range(4):
    t0 = time.perf_counter(); stage1(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    print("stage1 step", i, f"{dt:7.2f} ms  allocated {torch.cuda.memory_allocated()/1e9:6.2f} GB (base {base/1e9:.2f})  reserved {torch.cuda.memory_reserved()/1e9:6.2f} GB", flush=True)
gc.enable()

# stage 2 (the NoFs as modules on correspondence points) and the joint step through the fused loss partials (_loss_target)
del nerfs1, rays1, bg1, gt1
torch.cuda.empty_cache()
from moco_flow_amd import losses
B = 200000
nofs2 = [load(M.NoF(4, 128, 33, [2], "ind", 33, True), synth.nof_state(0, use_quat=True, tag=t, head_scale=0.25)) for t in ("bw", "fw")]
exyz, eind = M.Embedding(3, 5), M.Embedding(1, 16)
q, c2 = torch.randn(B, 3, device=dev) * 0.5, torch.randn(B, 3, device=dev) * 0.5
ind = torch.full((B, 1), -0.4, device=dev)
def stage2():
    for m in nofs2:
        m.zero_grad(set_to_none=True)
    loss = 0
    for m, src, dst in ((nofs2[0], q, c2), (nofs2[1], c2, q)):
        with torch.no_grad():
            inp = torch.cat([exyz(src), eind(ind)], -1)
        loss = loss + torch.nn.functional.mse_loss(m(inp, src), dst)
    loss.backward()
nerfs, nofs, rays, bg, gt, embs, kw = bench.joint_stage_setup(M, synth, torch, dev, 1024)
def joint_fused():
    for m in nerfs + nofs:
        m.zero_grad(set_to_none=True)
    res = M.render_rays(rays, bg, embs, nerfs, _loss_target=gt, **kw)
    t = losses.from_partials(res["loss_partials"])
    (t["img_loss"] + 0.1 * t["nof_local"] + 0.1 * t["nof_global"]).backward()
for name, f in (("stage2", stage2), ("joint, fused loss", joint_fused)):
    f(); f(); torch.cuda.synchronize(); gc.collect()
    base = torch.cuda.memory_allocated()
    gc.disable()
    for i in range(3):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
        print(name, "step", i, f"{dt:7.2f} ms  allocated {torch.cuda.memory_allocated()/1e9:6.2f} GB (base {base/1e9:.2f})", flush=True)
    gc.enable()
