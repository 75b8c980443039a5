"""Phase timeline of the fused render kernels (timing tool; needs a library built with -DMF_TIMELINE:
make -C moco_flow_amd/csrc EXTRA=-DMF_TIMELINE OUT=$PWD/build/ab/lib_tl.so, run with
MOCOFLOW_HIP_LIB=build/ab/lib_tl.so python tools/timeline.py C2|C2b|C3|C3g).  Prints, for waves 0 and 4 of workgroup 0, the shader-clock deltas between the
stamps of csrc/mf_render_bf16.hip / mf_bf16.hpp (tags: 1 tile start, 2 rays loaded, 3 NoF chain done, 4 encoded,
10+l trunk layer l done, 30 sigma head, 31 final layer, 32 extra operands, 33 extra layer, 5 tile end, 6 barrier,
7 composite, 8 barrier)."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import bench
import moco_flow_amd as M
from moco_flow_amd import rendering as R, synth

cfgname = sys.argv[1] if len(sys.argv) > 1 else "C2b"
cfg = bench.CONFIGS[cfgname]
dev = torch.device("cuda:0")
N, S = cfg["rays"], cfg["S"]
models = bench.build_models(M, synth, dev, cfg)
rays_np, bg_np = synth.rays(0, N, chained=(cfg["nof"] == "global"))
rays, bg = torch.from_numpy(rays_np).to(dev), torch.from_numpy(bg_np).to(dev)
z_steps = torch.linspace(0, 1, S, device=dev)
loc, glob = cfg["nof"] in ("local", "global"), cfg["nof"] == "global"
for it in range(3):
    out = R._render_pass(rays, bg, None, z_steps, False, None, 0, models["nerfs"][0], models["embs"], models["nofs"],
                         models["nof_embs"], loc, glob, False, True, precision=cfg["precision"])
torch.cuda.synchronize()
a = out["alphas"].flatten().cpu()
for w, base in ((0, 0), (4, 512)):
    prev = 0.0
    print(f"--- wave {w}")
    for k in range(250):
        tag, t = int(a[base + 2 * k]), float(a[base + 2 * k + 1])
        if k > 0 and (tag < 1 or tag > 40):
            break
        print(f"tag {tag:3d}  t {t:10.0f}  +{t - prev:8.0f}")
        prev = t

# every workgroup's start / end (chip-wide 100 MHz clock, 24 bits), written by the MF_TIMELINE build of the bf16 kernels
if True:
    G = 1
    tile = 256 if cfg["precision"] == "bf16" else 128
    while (G * S) % tile:
        G += 1
    ngroups = (N + G - 1) // G
    grid = min(ngroups, 256)
    al = out["alphas"].cpu()
    se = []
    for b in range(grid):
        lastg = b + ((ngroups - 1 - b) // grid) * grid
        row = al.flatten()[lastg * G * S:lastg * G * S + 4]
        se.append((float(row[2]), float(row[3])))
    se = np.array(se)
    t0 = se[:, 0].min()
    st, en = (se[:, 0] - t0) / 100.0, (se[:, 1] - t0) / 100.0          # microseconds
    print(f"workgroups: start  min {st.min():.2f} / median {np.median(st):.2f} / max {st.max():.2f} us;  end  min {en.min():.2f} / median {np.median(en):.2f} / max {en.max():.2f} us;"
          f"  run time min {(en - st).min():.2f} / median {np.median(en - st):.2f} / max {(en - st).max():.2f} us")
    order = np.argsort(en)
    print("latest workgroups (id: start, end):", [(int(b), round(float(st[b]), 1), round(float(en[b]), 1)) for b in order[-8:]])
    xcd = np.arange(grid) % 8
    for x in range(8):
        print(f"  XCD {x}: start median {np.median(st[xcd == x]):6.2f}  end median {np.median(en[xcd == x]):7.2f}  end max {en[xcd == x].max():7.2f}")
