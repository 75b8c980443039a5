"""Report (GPU): the three precision modes on BASELINE config C3's batch (both weight draws) and on C2's: PSNR-equivalent,
l2-rel and max-rel of every per-ray output against the fp32 ORACLE (CPU), and the fused launch's time.
    python tools/x3_report.py [n_rays]
"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import moco_flow_amd as M
from moco_flow_amd import rendering
from oracle import cpu_ref as R
import test_gpu_parity as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for name in ("r_nerf_dir_dense", "r_moco_local", "r_moco_global"):
    for draw, tags in (("bench", T.BENCH_TAGS), ("case", None)):
        if name == "r_nerf_dir_dense" and draw == "bench":
            continue
        for prec in ("f32", "bf16", "bf16x3"):
            c, res, want = T._full_size_case(M, R, name, n, prec, tags=tags)
            out = []
            for k in ("rgb_coarse", "depth_coarse", "opacity_coarse"):
                a, b = res[k].double().cpu(), want[k].double()
                mr = float(((a - b).abs() / (b.abs() + 1e-3)).max())
                out.append(f"{k.split('_')[0]} {T._psnr(res[k], want[k]):6.1f} dB l2 {T._l2rel(res[k], want[k]):.1e} maxrel {mr:.1e}")
            # time of the whole render_rays call in this mode (same batch)
            from helpers import build_case
            from cases import RENDER_CASES
            from moco_flow_amd import synth
            cc = dict(RENDER_CASES[name])
            rays_np, bg_np = synth.rays(0, n, chained=(cc.get("nof") == "global"))
            rays, bg = torch.from_numpy(rays_np).cuda(), torch.from_numpy(bg_np).cuda()
            embs, nerfs, kw = build_case(M, cc, 0, device="cuda", tags=tags)
            rendering.set_precision(prec)
            with torch.no_grad():
                for _ in range(3):
                    M.render_rays(rays, bg, embs, nerfs, **kw)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(10):
                    M.render_rays(rays, bg, embs, nerfs, **kw)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) * 100
            rendering.set_precision("f32")
            print(f"{name:18s} {draw:5s} {prec:7s} {ms:6.2f} ms/call | " + " | ".join(out), flush=True)
