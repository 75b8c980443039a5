"""GPU tool: how far each arithmetic mode of the HIP render pass sits from (a) the fp32 oracle and (b) the oracle of its
own arithmetic (oracle/bf16_ref.py), on BASELINE configs C3 (4096 x 64, both weight draws) and the C5 shard.

    python tools/arith_report.py [--rays 4096] [--modes bf16,bf16x3] [--arith bf16x3=bf16x3,bf16x3_r3] [--cache DIR]

``--arith mode=a,b``: which bf16_ref.ARITH records the mode is compared with (default: its own name).  Oracle results are
cached under --cache (they cost 5-30 s each), so several library builds (MOCOFLOW_HIP_LIB=...) can be compared cheaply.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from cases import RENDER_CASES          # noqa: E402
from helpers import build_case, relerr  # noqa: E402
import moco_flow_amd as M               # noqa: E402
from moco_flow_amd import rendering, synth   # noqa: E402
from oracle import bf16_ref as B        # noqa: E402
from oracle import cpu_ref as R         # noqa: E402

BENCH_TAGS = dict(coarse="nerf", fine="nerf_fine")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--modes", default="bf16,bf16x3")
    ap.add_argument("--arith", action="append", default=[])
    ap.add_argument("--cache", default=os.path.join(ROOT, "gpurun_out", "arith_cache"))
    ap.add_argument("--cases", default="r_moco_local,r_moco_global_fine")
    a = ap.parse_args()
    os.makedirs(a.cache, exist_ok=True)
    torch.set_num_threads(min(os.cpu_count() or 8, 32))
    versus = {m: [m] for m in a.modes.split(",")}
    for spec in a.arith:
        m, _, names = spec.partition("=")
        versus[m] = names.split(",")
    for case in a.cases.split(","):
        c = dict(RENDER_CASES[case])
        n = a.rays if c["M"] == 0 else a.rays // 4
        for draw, tags in (("bench", BENCH_TAGS), ("case", None)):
            rays_np, bg_np = synth.rays(0, n, chained=(c.get("nof") == "global"))
            rays, bg = torch.from_numpy(rays_np), torch.from_numpy(bg_np)
            for mode in a.modes.split(","):
                embs, nerfs, kw = build_case(M, c, 0, device="cuda", tags=tags)
                cap = {}
                try:
                    rendering.set_precision(mode)
                    with torch.no_grad():
                        res = M.render_rays(rays.cuda(), bg.cuda(), embs, nerfs, _capture=cap, **kw)
                finally:
                    rendering.set_precision("f32")
                res = {k: (v.materialize() if hasattr(v, "materialize") else v).cpu() for k, v in res.items()}
                extra = dict(_z_fine_override=cap["z_fine"].cpu()) if c["M"] > 0 else {}
                for name in ["f32"] + versus[mode]:
                    # (with a fine pass the oracle runs on THIS mode's fine depths, so the cache key carries the mode)
                    key = os.path.join(a.cache, f"{case}_{draw}_{n}_{name}" + (f"_z{mode}_{os.path.basename(os.environ.get('MOCOFLOW_HIP_LIB', 'lib'))}" if c["M"] > 0 else "") + ".pt")
                    if os.path.exists(key):
                        want = torch.load(key)
                    else:
                        e_o, n_o, kw_o = build_case(B.Backend(B.ARITH[name]), c, 0, tags=tags)
                        with torch.no_grad():
                            want = R.render_rays(rays, bg, e_o, n_o, **extra, **kw_o)
                        torch.save(want, key)
                    for tag in (["coarse", "fine"] if c["M"] > 0 else ["coarse"]):
                        keys = [f"rgb_{tag}", f"depth_{tag}", f"opacity_{tag}"]
                        print(f"{case:18s} {draw:5s} hip[{mode:6s}] vs oracle[{name:10s}] {tag:6s} {B.psnr_equiv(res[keys[0]], want[keys[0]]):6.1f} dB  max-rel "
                              + " ".join(f"{relerr(res[k], want[k]):.1e}" for k in keys) + "  l2 "
                              + " ".join(f"{B.l2rel(res[k], want[k]):.1e}" for k in keys), flush=True)


if __name__ == "__main__":
    main()
