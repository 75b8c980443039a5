#!/usr/bin/env python3
"""Time one training step of the joint MoCo-Flow stage (c2f.yaml shape: N_rand = 1024 rays x (128 coarse
+ 256 fine) samples, backward NoF -> NeRF(ind) with local + global consensus chains, trainer_moco_flow.py
:200-216, 317-328) through the drop-in (HIP forward + HIP backward) and with everything in
PyTorch-ROCm eager ops (tools/eager_ref.py: what the reference itself would run on this GPU)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import moco_flow_amd as M
import eager_ref as E      # tools/eager_ref.py
from moco_flow_amd import synth, rendering
from moco_flow_amd import autograd as _A
if os.environ.get("MF_NO_FUSED_MEAN") == "1":      # A/B: the consensus means through torch ops on the per-sample planes (rounds 2-3)
    from moco_flow_amd import rendering as _R
    _R.FUSED_CONSENSUS_MEAN = False
if os.environ.get("MF_WGRAD"):
    _A.set_wgrad_precision(os.environ["MF_WGRAD"])      # f32 | bf16x3
rendering.STRICT_RNG = False
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
S, Mi = 128, 128
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
crit = M.get_loss(dict(type="MSE"))
kw = dict(nof_embeddings=nof_embs, nof_models=nofs, chain_local=True, chain_global=True, N_samples=S,
          N_importance=Mi, noise_std=0, perturb=1.0)
mods = nerfs + nofs


def timeit(f, n=12):
    """median of n individually timed calls (a call that makes the caching allocator go to the driver -- the compacted
    consensus vectors change length from step to step -- costs tens of ms and would dominate a short mean)"""
    f(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t = time.perf_counter()
        f()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def fwd_only():
    with torch.no_grad():
        M.render_rays(rays, bg, embs, nerfs, **kw)


def fwd_bwd():
    for m in mods:
        m.zero_grad(set_to_none=True)
    res = M.render_rays(rays, bg, embs, nerfs, **kw)
    loss = crit(res, gt)
    for k in ("nof_local_disp_coarse", "nof_global_disp_coarse", "nof_local_disp_fine", "nof_global_disp_fine"):
        loss = loss + 0.1 * res[k].mean()            # consensus terms, trainer_moco_flow.py:317-328
    loss.backward()


def fwd_bwd_fast():
    """the trainer's fast path: the loss terms from the 12 (sum, count) partials, no compaction, no sync"""
    from moco_flow_amd import losses
    for m in mods:
        m.zero_grad(set_to_none=True)
    res = M.render_rays(rays, bg, embs, nerfs, _loss_target=gt, **kw)
    t = losses.from_partials(res["loss_partials"])
    (t["img_loss"] + 0.1 * (t["nof_local"] + t["nof_global"])).backward()


if os.environ.get("MF_TRAIN_FWD"):              # e.g. MF_TRAIN_FWD=bf16x3: the opt-in three-product training forward
    from moco_flow_amd import rendering as _r
    _r.set_train_forward_precision(os.environ["MF_TRAIN_FWD"])
if os.environ.get("MF_ONLY") == "fast":       # profiling: only the trainer's fast path (loss from the fused partials)
    print(f"  HIP forward + backward, loss from the fused partials : {timeit(fwd_bwd_fast):8.2f} ms")
    sys.exit(0)
if os.environ.get("MF_ONLY") == "step":       # profiling: only the shipped training step
    print(f"  HIP forward + backward (shipped) : {timeit(fwd_bwd):8.2f} ms")
    sys.exit(0)
print(f"N_rand={N} rays x ({S} + {S+Mi}) samples = {N*(2*S+Mi)/1e6:.2f} M samples/step, bw NoF + local + global chains")
print(f"  HIP forward (no_grad)            : {timeit(fwd_only):8.2f} ms")
print(f"  HIP forward + backward (shipped) : {timeit(fwd_bwd):8.2f} ms")
print(f"  same, loss from the fused partials : {timeit(fwd_bwd_fast):8.2f} ms")
def eager_fwd_bwd():
    for m in mods:
        m.zero_grad(set_to_none=True)
    res = E.render_rays_eager(rays, bg, embs, nerfs, nof_embs, nofs, True, True, N_samples=S, N_importance=Mi, perturb=1.0)
    loss = crit(res, gt)
    for k in ("nof_local_disp_coarse", "nof_global_disp_coarse", "nof_local_disp_fine", "nof_global_disp_fine"):
        loss = loss + 0.1 * res[k].mean()
    loss.backward()


if os.environ.get("MF_ONLY") != "hipbwd":
    print(f"  PyTorch-ROCm eager fwd+bwd       : {timeit(eager_fwd_bwd, n=5):8.2f} ms")
