"""Joint-stage / stage-1 training step, default (fp32 forward) vs the opt-in three-product forward: median ms per step
(bench.py's train_shape_legs alone)."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
import moco_flow_amd as M
from moco_flow_amd import rendering, synth
rendering.STRICT_RNG = False
out = bench.train_shape_legs(M, synth, torch, torch.device("cuda:0"), steps=8)
print(json.dumps({k: (round(v["ms_per_step"], 3) if isinstance(v, dict) else v) for k, v in out.items()}))
