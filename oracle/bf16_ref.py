"""TEST INFRASTRUCTURE ONLY -- the oracle with ROUNDING HOOKS at exactly the places the bf16 kernels round.

``oracle/cpu_ref.py`` restates the reference's fp32 path; the bf16 modes of the fused pass (BASELINE configs C3-C5,
``csrc/mf_bf16.hpp`` / ``csrc/mf_render_bf16.hip``) deliberately compute something else: bf16 matrix operands with fp32
accumulation.  Compared with the fp32 oracle alone they can only be held to "as far away as bf16 rounding puts them" --
a dropped k-range or a wrong split position that costs 3 dB would pass.  This module is the oracle OF THAT ARITHMETIC:
the same networks (subclasses of ``cpu_ref``'s containers, same state_dict keys, same call signatures, so ``cpu_ref``'s
``render_rays`` / ``nof_inference`` / ``nerf_inference`` drive them unchanged) with every operand rounded where the kernel
rounds it.  The reference for what the outputs mean is unchanged: /root/reference/models/nerf.py:78-102,
models/nof.py:69-82, models/embedding.py:42-47, models/rendering.py:49-83,121-192.

Pinning: with the hooks off (``F32``) every call degenerates to the very ``F.linear`` / ``torch.sin`` calls of
``cpu_ref`` -- ``tests/test_oracle_golden.py::test_bf16_ref_hooks_off_is_cpu_ref`` asserts ``torch.equal`` -- so this
module inherits ``cpu_ref``'s pinning to the reference-generated fixtures.  The hooks themselves restate OUR kernels'
choices, not anything in the reference (which has no reduced-precision path).

What is modelled (``Arith`` fields; kernel site in brackets):
  * matrix operands: "plain" = RNE to bf16; "split" = x = hi + lo, hi = bf16(x), lo = bf16(x - hi), three products
    hi*hi + lo*hi + hi*lo; "split3" = hi + mid + lo with the six products down to 2^-24; "hsplit*" = IEEE-half (hi, lo)
    pairs, three products (the NoF under bf16x3 since round 5: "hsplit_g", both operands at 2^5 x); "f32" = untouched
    [out_tile / mma_tile_x, split_operands, pack_operands, mf_pack.hip]
  * the NoF's image-index block as a per-ray fp32 bias, b + W[:, 33:66] emb(ind) by 33 sequential fp32 FMAs
    [nof_raybias_kernel]
  * activations: ReLU on the fp32 accumulator, then the operand rounding of the next layer [out_tile's epilogue, epi_step]
  * heads: "bf16act" = fp32 weights x bf16-rounded activations [valu_head, rounds 2-5]; "wsplit" on the NeRF = its sigma / rgb
    heads as matrix-pipe panels of (hi, lo) bf16 weight rows x bf16 activations [head_tile<K, 1>, round 6]; "f32acc" = fp32 weights x the fp32
    accumulators [epi_step NHEAD]; the NoF head "wsplit" = (Whi + Wlo) x bf16 activations [head_tile] or "split"
    [head_tile_x3]
  * sin / cos of the encodings: "exact" = OCML-class [sincosf]; "hw" = the transcendental unit's argument path,
    fract(fp32(angle * fp32(1 / 2 pi))) revolutions [sincos_rev]; "chain" = exact seeds + fp32 angle doublings
    [emb_eval with pow2 tables]
NOT modelled: the association order of the fp32 accumulation inside and between MFMAs (``acc="f64"`` accumulates the exact
products in float64 and rounds once: the "ideal" result every order is within ~1e-6 of); the transcendental unit's own
error behind its argument reduction; ``quat_transform<FAST>``'s v_rcp / v_sqrt (1 ulp each).  A kernel therefore sits
near this oracle but not on it: each accumulator that lands within ~1e-6 (relative) of a rounding boundary rounds the other
way (one operand ulp in one activation) -- see tests/test_gpu_parity.py for the measured distances.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, replace
from typing import Optional

import numpy as np
import torch
import torch.nn.functional as F

from . import cpu_ref as R
from .kornia_restated import quaternion_log_to_exp, quaternion_to_rotation_matrix


# ------------------------------------------------------------------ rounding primitives
def bf(x: torch.Tensor) -> torch.Tensor:
    """RNE to bfloat16, back in fp32 (v_cvt_pk_bf16_f32)."""
    return x.to(torch.bfloat16).to(torch.float32)


def split2(x):
    hi = bf(x)
    return hi, bf(x - hi)


def split3(x):
    hi = bf(x)
    mid = bf(x - hi)
    return hi, mid, bf(x - hi - mid)


def hf(x: torch.Tensor, ftz: bool = False) -> torch.Tensor:
    """RNE to IEEE half, back in fp32 (v_cvt_pk_f16_f32; |x| > 65504 -> inf).  ``ftz``: results below the normal range
    (|y| < 2^-14) flushed to zero -- what the split would see if the matrix unit dropped fp16 denormals (it does not:
    tools/proto/f16_denorm.hip; kept as a what-if)."""
    y = x.to(torch.float16).to(torch.float32)
    if ftz:
        y = torch.where(y.abs() < 2.0 ** -14, torch.zeros_like(y), y)
    return y


def hsplit2(x, scale_lo: float = 1.0, ftz: bool = False):
    """x = hi + lo in IEEE half: hi = half(x), lo = half((x - hi) * scale_lo) / scale_lo (22 significand bits; x - hi is
    exact in fp32).  scale_lo = 2^11 moves the lo term into hi's binade (a separate accumulator on the device)."""
    hi = hf(x, ftz)
    return hi, hf((x - hi) * scale_lo, ftz) / scale_lo


def operand_terms(x: torch.Tensor, W: torch.Tensor, how: str):
    """[(X_i, W_i)]: the products sum_i X_i W_i^T the matrix pipe evaluates for ``x W^T`` under arithmetic ``how``."""
    if how == "f32":
        return [(x, W)]
    if how == "plain":
        return [(bf(x), bf(W))]
    if how == "split":                       # Whi xhi + Wlo xhi + Whi xlo   (the lo*lo term is dropped)
        xh, xl = split2(x)
        Wh, Wl = split2(W)
        return [(xh, Wh), (xh, Wl), (xl, Wh)]
    if how == "wsplit":                      # (Whi + Wlo) x bf16(x)
        Wh, Wl = split2(W)
        xh = bf(x)
        return [(xh, Wh), (xh, Wl)]
    if how == "split3":                      # three terms each, the six products down to 2^-24
        xh, xm, xl = split3(x)
        Wh, Wm, Wl = split3(W)
        return [(xh, Wh), (xh, Wm), (xm, Wh), (xh, Wl), (xm, Wm), (xl, Wh)]
    if how in ("hsplit", "hsplit_s", "hsplit_ftz", "hsplit_g", "hsplit_gftz"):
        # fp16 (hi, lo) pairs, three products (v_mfma_f32_32x32x16_f16): 22 significand bits per operand.  _s: the lo terms scaled by
        # 2^11 (their own accumulator); _g: BOTH operands carried at 2^5 x their value (exact: powers of two; the accumulators
        # are 2^10 x, un-scaled in the epilogue), which keeps the lo terms of O(0.01 .. 1) values out of the denormal range;
        # ftz: what a matrix unit that flushed fp16 denormals would compute
        sc, ftz = (2.0 ** 11 if how == "hsplit_s" else 1.0), how.endswith("ftz")
        g = 32.0 if how in ("hsplit_g", "hsplit_gftz") else 1.0
        xh, xl = hsplit2(x * g, sc, ftz)
        Wh, Wl = hsplit2(W * g, sc, ftz)
        return [(xh / g, Wh / g), (xh / g, Wl / g), (xl / g, Wh / g)]
    raise ValueError(how)


def mm(terms, bias: Optional[torch.Tensor], start: Optional[torch.Tensor], acc: str) -> torch.Tensor:
    """start + bias + sum of the terms' products.  acc = "f32": ONE F.linear over the k-concatenated terms (with a single
    fp32 term this is cpu_ref's own call); "f64": exact products, float64 accumulation, one rounding to fp32."""
    X = terms[0][0] if len(terms) == 1 else torch.cat([t[0] for t in terms], -1)
    W = terms[0][1] if len(terms) == 1 else torch.cat([t[1] for t in terms], -1)
    if acc == "f32":
        y = F.linear(X, W, bias)
        return y if start is None else y + start
    y = X.double() @ W.double().t()
    if bias is not None:
        y = y + bias.double()
    if start is not None:
        y = y + start.double()
    return y.float()


def fma32(a, b, c):
    """fp32 fused multiply-add on fp32 tensors (the product of two fp32 is exact in float64)."""
    return (a.double() * b.double() + c.double()).float()


# ------------------------------------------------------------------ the arithmetic of a mode
@dataclass(frozen=True)
class Arith:
    name: str = "f32"
    acc: str = "f32"                 # "f32" | "f64" (see mm)
    # NoF (csrc/mf_bf16.hpp nof_eval / nof_eval_x3)
    nof_xyz: str = "f32"             # operands of the embedded xyz block (33 columns) of layer 0 / skip layers
    nof_ind_bias: bool = False       # image-index block as the per-ray fp32 bias (else: part of the embedded block)
    nof_hidden: str = "f32"
    nof_head: str = "f32"
    nof_xyz_sincos: str = "exact"
    # NeRF (nerf_eval / nerf_eval_x3)
    nerf_emb: str = "f32"            # operands of the embedded input (layer 0, skip layers) and of the extra block
    nerf_hidden: str = "f32"
    nerf_tail: str = "f32"           # xyz_encoding_final and extra_encoding
    nerf_heads: str = "f32"          # "f32" | "bf16act" (rounds 2-5's VALU heads) | "wsplit" (round 6: head panels) | "f32acc"
    nerf_xyz_sincos: str = "exact"
    nerf_extra_sincos: str = "exact"


F32 = Arith()
# set_precision("bf16") -- BASELINE configs C3-C5 as written ("MFMA bf16 hidden GEMMs")
BF16 = Arith(name="bf16", acc="f64", nof_xyz="split", nof_ind_bias=True, nof_hidden="plain", nof_head="wsplit",
             nof_xyz_sincos="hw", nerf_emb="plain", nerf_hidden="plain", nerf_tail="plain", nerf_heads="wsplit",
             nerf_xyz_sincos="hw", nerf_extra_sincos="hw")
# set_precision("bf16x3") as shipped at the end of round 3 (NoF xyz block from the transcendental unit)
BF16X3_R3 = Arith(name="bf16x3_r3", acc="f64", nof_xyz="split", nof_ind_bias=True, nof_hidden="split", nof_head="split",
                  nof_xyz_sincos="hw", nerf_emb="split", nerf_hidden="split", nerf_tail="split", nerf_heads="f32acc",
                  nerf_xyz_sincos="chain", nerf_extra_sincos="hw")
# round 4's set_precision("bf16x3"): the NoF in THREE-term bf16 operands (hi, mid, lo; six products per k-step, 24 mantissa bits)
# with its xyz block from exact seeds + doubling chains -- its output point feeds sin(512 x); the NeRF as before
BF16X3_R4 = replace(BF16X3_R3, name="bf16x3_r4", nof_xyz="split3", nof_hidden="split3", nof_head="split3", nof_xyz_sincos="chain")
# set_precision("bf16x3") as shipped (round 5): the NoF in IEEE-half (hi, lo) pairs at 2^5 x -- THREE products per k-step on
# v_mfma_f32_32x32x16_f16, 22 significand bits (csrc/mf_core.hpp kNofHalfX3): the same distance to the fp32 oracle as the
# three-term bf16 split at half the matrix instructions
BF16X3 = replace(BF16X3_R4, name="bf16x3", nof_xyz="hsplit_g", nof_hidden="hsplit_g", nof_head="hsplit_g")
# (the intermediate step: round 3's two-term NoF with the exact seeds)
BF16X3_2T = replace(BF16X3_R3, name="bf16x3_2t", nof_xyz_sincos="chain")

# (priced on the way: the same pairs without the 2^5 scale -- the lo terms of O(0.1) values are denormal halves: 109 dB / 3.9e-5)
BF16X3_H = replace(BF16X3, name="bf16x3_h", nof_xyz="hsplit", nof_hidden="hsplit", nof_head="hsplit")

ARITH = {"f32": F32, "bf16": BF16, "bf16x3": BF16X3, "bf16x3_2t": BF16X3_2T, "bf16x3_r3": BF16X3_R3, "bf16x3_r4": BF16X3_R4,
         "bf16x3_h": BF16X3_H}


# ------------------------------------------------------------------ E with the kernels' sin / cos
_INV_2PI_F32 = np.float32(0.15915494309189535)


def sincos_hw(arg: torch.Tensor):
    """sincos_rev (mf_bf16.hpp): revolutions = fp32(arg * fp32(1 / 2 pi)), v_fract_f32 (exact), then the unit's
    sin / cos of that fraction -- modelled as exact functions of the fp32 fraction."""
    rev = arg * torch.tensor(_INV_2PI_F32)
    fr = (rev - torch.floor(rev)).double()
    return torch.sin(2 * math.pi * fr).float(), torch.cos(2 * math.pi * fr).float()


class Embedding(R.Embedding):
    """cpu_ref.Embedding whose sin / cos come from the chosen unit (``mode``: "exact" | "hw" | "chain")."""
    mode = "exact"

    def _pow2(self):
        return all(float(f) == float(1 << k) for k, f in enumerate(self.freq_bands))

    def __call__(self, x):
        mode = self.mode
        if mode == "chain" and not self._pow2():
            mode = "exact"                                   # emb_eval: other tables take the direct (exact) path
        if mode == "exact" or self.N_freqs == 0:
            return R.Embedding.__call__(self, x)
        C = self.in_channels
        sn, cs = {}, {}
        for f, fr in enumerate(self.freq_bands):
            for c in range(C):
                arg = fr * x[:, c]
                if mode == "hw":
                    sn[f, c], cs[f, c] = sincos_hw(arg)
                    continue
                # emb_eval's chains: pair p = f C + c sits in lane-half pair pi = p // 2, entry m = pi // C of its chain;
                # every third entry is an exact seed, the others are two angle doublings from (f - 2, c)
                m = ((f * C + c) // 2) // C
                if m % 3 == 0:
                    sn[f, c], cs[f, c] = torch.sin(arg), torch.cos(arg)
                else:
                    s, co = sn[f - 2, c], cs[f - 2, c]
                    for _ in range(2):
                        t = s + s
                        s2 = t * co
                        co = fma32(-t, s, torch.ones_like(s))
                        s = s2
                    sn[f, c], cs[f, c] = s, co
        pieces = [x]
        for f, w in enumerate(self.weights):
            pieces.append(w * torch.stack([sn[f, c] for c in range(C)], -1))
            pieces.append(w * torch.stack([cs[f, c] for c in range(C)], -1))
        return torch.cat(pieces, -1)


# ------------------------------------------------------------------ N
class NeRF(R.NeRF):
    arith = F32

    def __call__(self, inputs, sigma_only=False, img_ind=None):
        a, p = self.arith, self.p
        cx = self.in_channels_xyz
        if not sigma_only:
            input_xyz, extra = torch.split(inputs, [cx, self.extra_feat_dim], dim=-1)
        else:
            input_xyz = inputs
        h = input_xyz
        acc = None
        for i in range(self.D):
            W, b = p[f"xyz_encoding_{i+1}.0.weight"], p[f"xyz_encoding_{i+1}.0.bias"]
            if i == 0:
                terms = operand_terms(input_xyz, W, a.nerf_emb)
            elif i in self.skips:
                if a.nerf_emb == a.nerf_hidden == "f32":
                    terms = operand_terms(torch.cat([input_xyz, h], -1), W, "f32")
                else:
                    terms = operand_terms(input_xyz, W[:, :cx], a.nerf_emb) + operand_terms(h, W[:, cx:], a.nerf_hidden)
            else:
                terms = operand_terms(h, W, a.nerf_hidden)
            acc = mm(terms, b, None, a.acc)
            h = F.relu(acc)                               # fp32 accumulators; rounded where the next layer consumes them
        def head(x, W, b):
            if a.nerf_heads == "wsplit":                   # matrix-pipe head panel: (Whi + Wlo) x bf16 activations [head_tile<.., 1>]
                return mm(operand_terms(x, W, "wsplit"), b, None, a.acc)
            return mm([(bf(x) if a.nerf_heads == "bf16act" else x, W)], b, None, a.acc)
        sigma = head(h, p["sigma.weight"], p["sigma.bias"])
        if sigma_only:
            return sigma
        feat = mm(operand_terms(h, p["xyz_encoding_final.weight"], a.nerf_tail), p["xyz_encoding_final.bias"], None, a.acc)
        if self.extra_feat_type == "latent_code":
            raise NotImplementedError("NeRF model does not support latent code yet!!!")
        We, be = p["extra_encoding.0.weight"], p["extra_encoding.0.bias"]
        if a.nerf_tail == a.nerf_emb == "f32":
            terms = operand_terms(torch.cat([feat, extra], -1), We, "f32")
        else:
            terms = operand_terms(feat, We[:, :self.W], a.nerf_tail) + \
                (operand_terms(extra, We[:, self.W:], a.nerf_emb) if self.extra_feat_dim > 0 else [])
        e = F.relu(mm(terms, be, None, a.acc))
        rgb = torch.sigmoid(head(e, p["rgb.0.weight"], p["rgb.0.bias"]))
        return torch.cat([rgb, sigma], -1)


# ------------------------------------------------------------------ F
class NoF(R.NoF):
    arith = F32

    def _ray_bias(self, W_ind, b, ind_e):
        """nof_raybias_kernel: acc = b; acc = fma(W[:, 33 + k], e[k], acc) for k = 0..32, one row per distinct index."""
        uniq, inv = torch.unique(ind_e, dim=0, return_inverse=True)
        acc = b.unsqueeze(0).expand(uniq.shape[0], -1).contiguous()
        for k in range(W_ind.shape[1]):
            acc = fma32(W_ind[:, k].unsqueeze(0), uniq[:, k:k + 1], acc)
        return acc[inv]

    def __call__(self, inputs, xyz, img_ind=None):
        if self.extra_feat_type == "latent_code":
            raise NotImplementedError("NoF model does not support latent code yet!!!")
        a, p = self.arith, self.p
        cx = self.in_channels_xyz
        cin = inputs.shape[1]
        x_e, ind_e = inputs[:, :cx], inputs[:, cx:]
        u = None
        for i in range(self.D):
            W, b = p[f"nof_encoding_{i+1}.0.weight"], p[f"nof_encoding_{i+1}.0.bias"]
            emb_here = i == 0 or i in self.skips
            bias, start = b, None
            if emb_here and a.nof_ind_bias:                 # xyz block on the matrix pipe, index block in the accumulators' start
                e_terms = operand_terms(x_e, W[:, :cx], a.nof_xyz)
                start, bias = self._ray_bias(W[:, cx:cin], b, ind_e), None
            elif emb_here and i > 0 and a.nof_xyz == a.nof_hidden == "f32":
                e_terms = None                               # one GEMM over cat([inputs, u]) below: cpu_ref's own call
            elif emb_here:
                e_terms = operand_terms(inputs, W[:, :cin], a.nof_xyz)
            if i == 0:
                terms = e_terms
            elif emb_here and e_terms is None:
                terms = operand_terms(torch.cat([inputs, u], -1), W, "f32")
            elif emb_here:
                terms = e_terms + operand_terms(u, W[:, cin:], a.nof_hidden)
            else:
                terms = operand_terms(u, W, a.nof_hidden)
            u = F.relu(mm(terms, bias, start, a.acc))
        head = mm(operand_terms(u, p["nof_encoding_final.weight"], a.nof_head), p["nof_encoding_final.bias"], None, a.acc)
        if self.use_quat:
            v, s, t = head[:, :3], head[:, 3:6], head[:, 6:9]
            r = quaternion_to_rotation_matrix(quaternion_log_to_exp(v))
            return torch.bmm((xyz - s).unsqueeze(1), r).squeeze(1) + s + t
        return head + xyz


# ------------------------------------------------------------------ building a model set of one arithmetic
class Backend:
    """What tests/helpers.build_case wants -- Embedding / NeRF / NoF constructors -- with every object it builds set to
    ``arith``.  Embeddings are told apart by their shape, as the kernels' four tables are: 3 channels + 10 (or fewer)
    frequencies next to a NeRF = xyz; (3, 4) = dir; (1, 2) = the NeRF's index; (3, 5) = NoF xyz; (1, 16) = NoF index."""

    def __init__(self, arith: Arith):
        self.arith = arith

    def Embedding(self, in_channels, N_freqs, logscale=True):
        e = Embedding(in_channels, N_freqs, logscale)
        a = self.arith
        if (in_channels, N_freqs) == (3, 5):
            e.mode = a.nof_xyz_sincos
        elif (in_channels, N_freqs) == (1, 16):
            e.mode = "exact"                                # nof_raybias_kernel: OCML sincosf
        elif (in_channels, N_freqs) in ((3, 4), (1, 2)):
            e.mode = a.nerf_extra_sincos
        else:
            e.mode = a.nerf_xyz_sincos
        return e

    def NeRF(self, *args, **kw):
        m = NeRF(*args, **kw)
        m.arith = self.arith
        return m

    def NoF(self, *args, **kw):
        m = NoF(*args, **kw)
        m.arith = self.arith
        return m


def psnr_equiv(a, b):
    """-10 log10 mean((a - b)^2) in float64 (models/metrics.py:4-13's formula)."""
    mse = float(((a.double() - b.double()) ** 2).mean())
    return -10 * math.log10(mse) if mse > 0 else 200.0


def l2rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
