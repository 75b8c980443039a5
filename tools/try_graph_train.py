"""Experiment: the joint-stage training step (forward + loss from the fused partials + backward, tools/time_moco_step.py's
fast path) captured in a HIP graph through torch.cuda.CUDAGraph and replayed -- wall time per step vs the eager step, and
that the replayed gradients equal the eager ones bit for bit."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import moco_flow_amd as M
from moco_flow_amd import synth, rendering, losses
rendering.STRICT_RNG = False
N, S, Mi = 1024, 128, 128
dev = torch.device("cuda")


def load(m, sd):
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); return m.to(dev)


nerfs = [load(M.NeRF(8, 256, 63, [4], "ind", 5), synth.nerf_state(0, extra_feat_type="ind", extra_feat_dim=5, regime="dense", tag=t)) for t in ("coarse", "fine")]
nofs = [load(M.NoF(4, 128, 33, [2], "ind", 33, True), synth.nof_state(0, use_quat=True, tag=t, head_scale=0.25)) for t in ("bw", "fw")]
embs = [M.Embedding(3, 10), M.Embedding(1, 2), None]
nof_embs = [M.Embedding(3, 5), M.Embedding(1, 16)]
r, b = synth.rays(0, N, chained=True)
rays, bg = torch.from_numpy(r).to(dev), torch.from_numpy(b).to(dev)
gt = torch.rand(N, 3, device=dev)
# perturb = 0: the random draws of a captured step would be frozen into the graph
kw = dict(nof_embeddings=nof_embs, nof_models=nofs, chain_local=True, chain_global=True, N_samples=S, N_importance=Mi, noise_std=0, perturb=0)
mods = nerfs + nofs
params = [p for m in mods for p in m.parameters()]


def step():
    res = M.render_rays(rays, bg, embs, nerfs, _loss_target=gt, **kw)
    t = losses.from_partials(res["loss_partials"])
    (t["img_loss"] + 0.1 * (t["nof_local"] + t["nof_global"])).backward()


def timeit(f, n=15):
    f(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        t = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    return sorted(ts)[len(ts) // 2]


def eager():
    for p in params:
        p.grad = None
    step()


print(f"eager step        : {timeit(eager):.2f} ms")
ref = [p.grad.clone() for p in params]
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        for p in params:
            p.grad = None
        step()
torch.cuda.current_stream().wait_stream(s)
for p in params:
    p.grad = None
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
g.replay(); torch.cuda.synchronize()
same = all(torch.equal(p.grad, q) for p, q in zip(params, ref))
print(f"graph replay step : {timeit(g.replay):.2f} ms   gradients identical to the eager step: {same}")
