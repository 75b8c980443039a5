"""Design tool (CPU, not product, not a test): what an arithmetic choice of the bf16 modes costs against the fp32 oracle,
on BASELINE config C3's batch and both weight draws, by running oracle/bf16_ref.py with modified ``Arith`` records.

    python tools/bf16_explore.py [n_rays] [variant ...]        (EMU_CASES=r_moco_local,r_moco_global_fine)

A variant is a name of oracle.bf16_ref.ARITH or "<base>:<field>=<value>,..." (e.g. "bf16x3:nof_xyz=split3,nof_hidden=f32").
(Successor of round 3's tools/bf16_emulate.py, whose emulation became the oracle of the bf16 arithmetic.)
"""
import os
import sys
from dataclasses import replace

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from cases import RENDER_CASES          # noqa: E402
from helpers import build_case, relerr  # noqa: E402
from moco_flow_amd import synth         # noqa: E402
from oracle import bf16_ref as B        # noqa: E402
from oracle import cpu_ref as R         # noqa: E402


def parse(spec):
    base, _, mods = spec.partition(":")
    a = B.ARITH[base]
    if mods:
        kv = {}
        for m in mods.split(","):
            k, v = m.split("=")
            kv[k] = (v == "True") if v in ("True", "False") else v
        a = replace(a, name=spec, **kv)
    return a


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    names = sys.argv[2:] or ["bf16", "bf16x3_r3", "bf16x3"]
    torch.set_num_threads(os.cpu_count() or 8)
    cases = (("r_moco_local", n), ("r_moco_global_fine", max(n // 4, 64)))
    if os.environ.get("EMU_CASES"):
        cases = tuple(cs for cs in cases if cs[0] in os.environ["EMU_CASES"].split(","))
    for case, n_case in cases:
        c = dict(RENDER_CASES[case])
        for draw, tags in (("bench", dict(coarse="nerf", fine="nerf_fine")), ("case", None)):
            rays_np, bg_np = synth.rays(0, n_case, chained=(c.get("nof") == "global"))
            rays, bg = torch.from_numpy(rays_np), torch.from_numpy(bg_np)
            embs, nerfs, kw = build_case(R, c, 0, tags=tags)
            cap = {}
            with torch.no_grad():
                want = R.render_rays(rays, bg, embs, nerfs, _capture=cap, **kw)
            for name in names:
                embs_e, nerfs_e, kw_e = build_case(B.Backend(parse(name)), c, 0, tags=tags)
                extra = dict(_z_fine_override=cap["z_fine"]) if c["M"] > 0 else {}
                with torch.no_grad():
                    got = R.render_rays(rays, bg, embs_e, nerfs_e, **extra, **kw_e)
                tag = "fine" if c["M"] > 0 else "coarse"
                keys = [f"rgb_{tag}", f"depth_{tag}", f"opacity_{tag}"]
                print(f"{case:20s} {draw:5s} {name:44s} {B.psnr_equiv(got[keys[0]], want[keys[0]]):6.1f} dB  max-rel "
                      + " ".join(f"{relerr(got[k], want[k]):.1e}" for k in keys) + "  l2 "
                      + " ".join(f"{B.l2rel(got[k], want[k]):.1e}" for k in keys), flush=True)


if __name__ == "__main__":
    main()
