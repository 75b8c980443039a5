#!/usr/bin/env python3
"""Time one stage-2 training step (trainer_nof.py:113-132: N_sampled = 100 000 inside + outside correspondence points
through the backward and the forward NoF called as modules, L2 losses) through the drop-in: HIP forward + HIP backward
(autograd.NofModule) vs the torch-recompute backward vs everything in PyTorch-ROCm eager ops."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import moco_flow_amd as M
import eager_ref as E   # tools/eager_ref.py
from moco_flow_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
dev = torch.device("cuda")
load = lambda m, sd: (m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}), m.to(dev))[1]
nofs = [load(M.NoF(4, 128, 33, [2], "ind", 33, True), synth.nof_state(0, use_quat=True, tag=t, head_scale=0.25)) for t in ("bw", "fw")]
exyz, eind = M.Embedding(3, 5), M.Embedding(1, 16)
q, c = torch.randn(B, 3, device=dev) * 0.5, torch.randn(B, 3, device=dev) * 0.5
ind = torch.full((B, 1), -0.4, device=dev)


def step(call):
    for m in nofs:
        m.zero_grad(set_to_none=True)
    loss = 0
    for m, src, dst in ((nofs[0], q, c), (nofs[1], c, q)):
        with torch.no_grad():
            inp = torch.cat([exyz(src), eind(ind)], -1)
        loss = loss + torch.nn.functional.mse_loss(call(m, inp, src), dst)
    loss.backward()


def timeit(f, n=5):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3


print(f"{B} correspondence points, bw + fw NoF, forward + backward")
print(f"  HIP forward + HIP backward (shipped)      : {timeit(lambda: step(lambda m, i, x: m(i, x))):7.2f} ms")
print(f"  PyTorch-ROCm eager (reference op sequence): {timeit(lambda: step(lambda m, i, x: E.nof_forward(m, i, x))):7.2f} ms")
