#!/usr/bin/env python3
"""Duration of the training forward alone (mf_render_pass with every dump plane) at the joint-stage shape:
1024 rays x 256 samples, bw NoF -> NeRF(ind), local + global chains, fp32.  Compared with the gradient-free pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import moco_flow_amd as M
from moco_flow_amd import synth, rendering, _lib as L
N, S = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024), 256
dev = torch.device("cuda")
load = lambda m, sd: (m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}), m.to(dev))[1]
nerf = load(M.NeRF(8, 256, 63, [4], "ind", 5), synth.nerf_state(0, extra_feat_type="ind", extra_feat_dim=5, regime="dense"))
nofs = [load(M.NoF(4, 128, 33, [2], "ind", 33, True), synth.nof_state(0, use_quat=True, tag=t, head_scale=0.25)) for t in ("bw", "fw")]
embs = [M.Embedding(3, 10), M.Embedding(1, 2), None]
nof_embs = [M.Embedding(3, 5), M.Embedding(1, 16)]
r, b = synth.rays(0, N, chained=True)
rays, bg = torch.from_numpy(r).to(dev), torch.from_numpy(b).to(dev)
t = torch.linspace(0, 1, S, device=dev)
z = (rays[:, 6:7] * (1 - t) + rays[:, 7:8] * t).contiguous()
# round 5: both arithmetics of the training forward (MF_PREC_F32 | MF_PREC_BF16X3), with and without the dump planes, and the
# chain-free pass (bw NoF -> NeRF only) beside the local + global one
for chains in (True, False):
    for prec in ("f32", "bf16x3"):
        for dump in (False, True):
            args = (rays, bg, z, None, False, None, L.MF_ACT_RELU, nerf, embs, nofs, nof_embs, chains, chains, False, True)
            with torch.no_grad():
                for _ in range(3):
                    rendering._render_pass(*args, dump=dump, precision=prec)
                ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(10)]
                for s, e in ev:
                    s.record(); rendering._render_pass(*args, dump=dump, precision=prec); e.record()
                torch.cuda.synchronize()
            ms = sorted(s.elapsed_time(e) for s, e in ev)
            print(f"chains={chains} prec={prec} dump={dump}: median {ms[len(ms)//2]:.3f} ms (min {ms[0]:.3f})", flush=True)
