#!/usr/bin/env python3
"""Time one training step of the stage-1 objective (trainer_nerf.py:149-169 shape: N_rand rays x
(128 coarse + 128 fine -> 256) samples) through the drop-in: HIP forward + HIP backward, the library-GEMM backward variant, and the
same pass done entirely with PyTorch-ROCm ops (what the reference itself would run on this GPU)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import moco_flow_amd as M
import eager_ref as E   # tools/eager_ref.py
from moco_flow_amd import synth, rendering
from moco_flow_amd import autograd as _A
if os.environ.get("MF_TRAIN_FWD"):
    rendering.set_train_forward_precision(os.environ["MF_TRAIN_FWD"])      # f32 | bf16x3
if os.environ.get("MF_DX"):
    _A.set_dx_precision(os.environ["MF_DX"])      # f32 | bf16x3
if os.environ.get("MF_WGRAD"):
    _A.set_wgrad_precision(os.environ["MF_WGRAD"])      # f32 | bf16x3
rendering.STRICT_RNG = False
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
S, Mi = 128, 128
dev = torch.device("cuda")
def mk(tag):
    m = M.NeRF(8, 256, 63, [4], "dir", 27); m.load_state_dict({k: torch.from_numpy(v) for k, v in synth.nerf_state(0, regime="dense", tag=tag).items()}); return m.to(dev)
nerfs = [mk("coarse"), mk("fine")]
embs = [M.Embedding(3, 10), None, M.Embedding(3, 4)]
r, b = synth.rays(0, N); rays, bg = torch.from_numpy(r).to(dev), torch.from_numpy(b).to(dev)
gt = torch.rand(N, 3, device=dev)
crit = M.get_loss(dict(type="MSE"))
kw = dict(N_samples=S, N_importance=Mi, noise_std=0, perturb=0)
def sync(): torch.cuda.synchronize()
def timeit(f, n=5):
    f(); sync(); t = time.perf_counter()
    for _ in range(n): f()
    sync(); return (time.perf_counter() - t) / n * 1e3
def fwd_only():
    with torch.no_grad(): M.render_rays(rays, bg, embs, nerfs, **kw)
def fwd_bwd():
    for m in nerfs: m.zero_grad(set_to_none=True)
    crit(M.render_rays(rays, bg, embs, nerfs, **kw), gt).backward()
def eager_fwd():
    with torch.no_grad():
        z = (rays[:, 6:7] * (1 - torch.linspace(0, 1, S, device=dev)) + rays[:, 7:8] * torch.linspace(0, 1, S, device=dev)).contiguous()
        c = E.render_pass(rays, bg, z, None, "relu", nerfs[0], embs, None, None, False, False, False, None)
        z2 = rendering.resample_merge(z, c["weights"], Mi)
        E.render_pass(rays, bg, z2, None, "relu", nerfs[1], embs, None, None, False, False, False, None)
if os.environ.get("MF_TRAIN_FWD"):              # e.g. MF_TRAIN_FWD=bf16x3: the opt-in three-product training forward
    from moco_flow_amd import rendering as _r
    _r.set_train_forward_precision(os.environ["MF_TRAIN_FWD"])
if os.environ.get("MF_ONLY") in ("hipbwd", "step"):      # profiling: only the shipped training path
    print(f"  HIP forward + HIP dX chain + dW GEMMs : {timeit(fwd_bwd):8.2f} ms")
    sys.exit(0)
print(f"N_rand={N} rays x ({S} + {S+Mi}) samples = {N*(2*S+Mi)/1e6:.2f} M samples/step")
print(f"  HIP forward (no_grad)            : {timeit(fwd_only):8.2f} ms")
print(f"  PyTorch-ROCm eager forward        : {timeit(eager_fwd):8.2f} ms")
print(f"  HIP forward + HIP dX chain + dW GEMMs : {timeit(fwd_bwd):8.2f} ms")
def eager_fwd_bwd():
    for m in nerfs: m.zero_grad(set_to_none=True)
    crit(E.render_rays_eager(rays, bg, embs, nerfs, N_samples=S, N_importance=Mi), gt).backward()
print(f"  PyTorch-ROCm eager fwd+bwd (tools/eager_ref.py) : {timeit(eager_fwd_bwd):8.2f} ms")
