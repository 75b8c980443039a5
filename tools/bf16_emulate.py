"""Design tool (CPU, not product, not a test): emulate the bf16 modes of the fused pass with torch ops on top of the
ORACLE's networks and measure what each arithmetic choice costs against the fp32 oracle on BASELINE config C3's batch.

    python tools/bf16_emulate.py [n_rays] [variant ...]

Emulated: operands rounded to bf16 (RNE) with fp32 accumulation; "split" = two-term bf16 split x = hi + lo with the
three products hi*hi + hi*lo + lo*hi (16 mantissa bits).  Not emulated: the accumulation order inside an MFMA, the
transcendental-unit sin / cos of the encodings (both orders of magnitude under bf16 rounding).

Variants (NoF / NeRF arithmetic):
  r2        round-2 kernel: NoF embedded input (xyz + image index) split, hidden plain, head weights split;
            NeRF everything plain
  bias      r2 with the image-index block as an exact fp32 per-ray bias (this round's fast mode)
  bias_h1   bias + head weights hi-only
  x3        bias + NoF hidden GEMMs and head as three-product splits
  x3s       x3 + the NeRF's last trunk layer and sigma head in split precision
  x3n       x3 + the NeRF's embedded input split (layer 0 and skip)
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

from cases import RENDER_CASES          # noqa: E402
from helpers import build_case          # noqa: E402
from moco_flow_amd import synth         # noqa: E402
from oracle import cpu_ref as R         # noqa: E402
from oracle.kornia_restated import quaternion_log_to_exp, quaternion_to_rotation_matrix   # noqa: E402


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def split(x):
    hi = bf(x)
    return hi, bf(x - hi)


def lin_plain(x, W):
    return F.linear(bf(x), bf(W))


def lin_split(x, W):
    xh, xl = split(x)
    Wh, Wl = split(W)
    return F.linear(xh, Wh) + F.linear(xl, Wh) + F.linear(xh, Wl)


class NoFEmu(R.NoF):
    opt = dict(ind_exact=False, head="split_w", hidden="plain")

    def __call__(self, inputs, xyz, img_ind=None):
        p, o = self.p, self.opt
        if o.get("exact"):
            return R.NoF.__call__(self, inputs, xyz, img_ind)
        cx = self.in_channels_xyz
        u = None
        for i in range(self.D):
            W, b = p[f"nof_encoding_{i+1}.0.weight"], p[f"nof_encoding_{i+1}.0.bias"]
            acc = b.unsqueeze(0)
            if i == 0 or i in self.skips:
                We = W[:, :inputs.shape[1]]
                if o["ind_exact"] and o.get("xyz") == "in_split":      # input split, weights hi-only: two products
                    xh, xl = split(inputs[:, :cx])
                    acc = acc + F.linear(xh, bf(We[:, :cx])) + F.linear(xl, bf(We[:, :cx])) + F.linear(inputs[:, cx:], We[:, cx:])
                elif o["ind_exact"] and o.get("xyz") == "w_split":     # weights split, input hi-only
                    Wa, Wb = split(We[:, :cx])
                    acc = acc + F.linear(bf(inputs[:, :cx]), Wa) + F.linear(bf(inputs[:, :cx]), Wb) + F.linear(inputs[:, cx:], We[:, cx:])
                elif o["ind_exact"]:
                    acc = acc + lin_split(inputs[:, :cx], We[:, :cx]) + F.linear(inputs[:, cx:], We[:, cx:])
                else:
                    acc = acc + lin_split(inputs, We)
                Wh = W[:, inputs.shape[1]:]
            else:
                Wh = W
            if i > 0:
                if o["hidden"] == "split" and i not in o.get("act_plain_layers", ()):
                    acc = acc + lin_split(u, Wh)
                elif o["hidden"] == "split":                  # this layer: weights split, activations plain
                    Wa, Wb = split(Wh)
                    acc = acc + F.linear(bf(u), Wa) + F.linear(bf(u), Wb)
                elif o["hidden"] == "wsplit":                 # weights split, activations plain: two products
                    Wa, Wb = split(Wh)
                    acc = acc + F.linear(bf(u), Wa) + F.linear(bf(u), Wb)
                else:
                    acc = acc + lin_plain(u, Wh)
            u = F.relu(acc)                                   # fp32 accumulators; rounded where consumed
        W, b = p["nof_encoding_final.weight"], p["nof_encoding_final.bias"]
        if o["head"] == "split_w":                            # r2: (Whi + Wlo) * bf16(hidden)
            Wh, Wl = split(W)
            head = F.linear(bf(u), Wh) + F.linear(bf(u), Wl) + b
        elif o["head"] == "hi":
            head = lin_plain(u, W) + b
        else:
            head = lin_split(u, W) + b
        if self.use_quat:
            v, s, t = head[:, :3], head[:, 3:6], head[:, 6:9]
            r = quaternion_to_rotation_matrix(quaternion_log_to_exp(v))
            return torch.bmm((xyz - s).unsqueeze(1), r).squeeze(1) + s + t
        return head + xyz


class NeRFEmu(R.NeRF):
    opt = dict(last="plain", emb="plain")

    def __call__(self, inputs, sigma_only=False, img_ind=None):
        p, o = self.p, self.opt
        if not sigma_only:
            input_xyz, extra = torch.split(inputs, [self.in_channels_xyz, self.extra_feat_dim], dim=-1)
        else:
            input_xyz = inputs
        emb_lin = lin_split if o["emb"] == "split" else lin_plain
        h = None
        for i in range(self.D):
            W, b = p[f"xyz_encoding_{i+1}.0.weight"], p[f"xyz_encoding_{i+1}.0.bias"]
            acc = b.unsqueeze(0)
            if i == 0 or i in self.skips:
                acc = acc + emb_lin(input_xyz, W[:, :self.in_channels_xyz])
                Wh = W[:, self.in_channels_xyz:]
            else:
                Wh = W
            if i > 0:
                last = i == self.D - 1
                if last and o["last"] == "split":
                    acc = acc + lin_split(h, Wh)
                elif last and o["last"] == "wsplit":          # weights split, activations plain: two products
                    Wa, Wb = split(Wh)
                    acc = acc + F.linear(bf(h), Wa) + F.linear(bf(h), Wb)
                elif o.get("hidden") == "split":             # every hidden layer in three products
                    acc = acc + lin_split(h, Wh)
                elif o.get("hidden") == "wsplit":
                    Wa, Wb = split(Wh)
                    acc = acc + F.linear(bf(h), Wa) + F.linear(bf(h), Wb)
                elif o.get("hidden") == "asplit":
                    ha, hb = split(h)
                    acc = acc + F.linear(ha, bf(Wh)) + F.linear(hb, bf(Wh))
                else:
                    acc = acc + lin_plain(h, Wh)
            h = F.relu(acc)
        if o["last"] in ("split", "wsplit", "sig32"):         # sigma head on the fp32 accumulators of the last layer
            sigma = F.linear(h, p["sigma.weight"], p["sigma.bias"])
        else:
            sigma = F.linear(bf(h), p["sigma.weight"], p["sigma.bias"])        # fp32 weights x bf16 activations
        if sigma_only:
            return sigma
        tail = lin_split if o.get("tail") == "split" else lin_plain
        feat = tail(h, p["xyz_encoding_final.weight"]) + p["xyz_encoding_final.bias"]
        e = F.relu(tail(torch.cat([feat, extra], -1), p["extra_encoding.0.weight"]) + p["extra_encoding.0.bias"])
        rgb = torch.sigmoid(F.linear(e if o.get("tail") == "split" else bf(e), p["rgb.0.weight"], p["rgb.0.bias"]))
        return torch.cat([rgb, sigma], -1)


VARIANTS = {
    "r2":      (dict(ind_exact=False, head="split_w", hidden="plain"), dict(last="plain", emb="plain")),
    "bias":    (dict(ind_exact=True, head="split_w", hidden="plain"), dict(last="plain", emb="plain")),
    "bias_h1": (dict(ind_exact=True, head="hi", hidden="plain"), dict(last="plain", emb="plain")),
    "x3":      (dict(ind_exact=True, head="split", hidden="split"), dict(last="plain", emb="plain")),
    "x3s":     (dict(ind_exact=True, head="split", hidden="split"), dict(last="split", emb="plain")),
    "x3n":     (dict(ind_exact=True, head="split", hidden="split"), dict(last="plain", emb="split")),
    "f32nof":  (dict(exact=True), dict(last="plain", emb="plain")),
    "f32nof_n": (dict(exact=True), dict(last="plain", emb="split")),
    "f32nof_sn": (dict(exact=True), dict(last="split", emb="split")),
    "x3_sig32":  (dict(ind_exact=True, head="split", hidden="split"), dict(last="sig32", emb="plain")),
    "x3_w7":     (dict(ind_exact=True, head="split", hidden="split"), dict(last="wsplit", emb="plain")),
    "x3n_sig32": (dict(ind_exact=True, head="split", hidden="split"), dict(last="sig32", emb="split")),
    "x3n_w7":    (dict(ind_exact=True, head="split", hidden="split"), dict(last="wsplit", emb="split")),
    "x3n_wall":  (dict(ind_exact=True, head="split", hidden="split"), dict(last="wsplit", emb="split", hidden="wsplit")),
    "x3n_aall":  (dict(ind_exact=True, head="split", hidden="split"), dict(last="sig32", emb="split", hidden="asplit")),
    "x2w":       (dict(ind_exact=True, head="split_w", hidden="wsplit"), dict(last="plain", emb="plain")),
    "x2w_n_w7":  (dict(ind_exact=True, head="split_w", hidden="wsplit"), dict(last="wsplit", emb="split")),
    "x3_l2p":   (dict(ind_exact=True, head="split", hidden="split", act_plain_layers=(2,)), dict(last="plain", emb="plain")),
    "x3_l12p":  (dict(ind_exact=True, head="split", hidden="split", act_plain_layers=(1, 2)), dict(last="plain", emb="plain")),
    "x3_l1p":   (dict(ind_exact=True, head="split", hidden="split", act_plain_layers=(1,)), dict(last="plain", emb="plain")),
    "bias_in2":  (dict(ind_exact=True, head="split_w", hidden="plain", xyz="in_split"), dict(last="plain", emb="plain")),
    "bias_w2":   (dict(ind_exact=True, head="split_w", hidden="plain", xyz="w_split"), dict(last="plain", emb="plain")),
    "x3full":  (dict(ind_exact=True, head="split", hidden="split"), dict(last="split", emb="split", hidden="split", tail="split")),
    "x3hid":   (dict(ind_exact=True, head="split", hidden="split"), dict(last="split", emb="split", hidden="split")),
    "x3sn":    (dict(ind_exact=True, head="split", hidden="split"), dict(last="split", emb="split")),
}


class Backend:
    Embedding = R.Embedding
    NeRF = NeRFEmu
    NoF = NoFEmu


def psnr(a, b):
    mse = float(((a.double() - b.double()) ** 2).mean())
    return -10 * np.log10(mse) if mse > 0 else 200.0


def l2rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    names = sys.argv[2:] or list(VARIANTS)
    torch.set_num_threads(8)
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
                NoFEmu.opt, NeRFEmu.opt = VARIANTS[name]
                embs_e, nerfs_e, kw_e = build_case(Backend, c, 0, tags=tags)
                extra = dict(_z_fine_override=cap["z_fine"]) if c["M"] > 0 else {}
                with torch.no_grad():
                    got = R.render_rays(rays, bg, embs_e, nerfs_e, **extra, **kw_e)
                tag = "fine" if c["M"] > 0 else "coarse"
                print(f"{case:20s} {draw:5s} {name:8s} rgb_{tag} {psnr(got['rgb_' + tag], want['rgb_' + tag]):5.1f} dB  l2 rgb "
                      f"{l2rel(got['rgb_' + tag], want['rgb_' + tag]):.1e} depth {l2rel(got['depth_' + tag], want['depth_' + tag]):.1e} "
                      f"opacity {l2rel(got['opacity_' + tag], want['opacity_' + tag]):.1e}"
                      + (f"  | rgb_coarse {psnr(got['rgb_coarse'], want['rgb_coarse']):5.1f} dB" if c["M"] > 0 else ""), flush=True)


if __name__ == "__main__":
    main()
