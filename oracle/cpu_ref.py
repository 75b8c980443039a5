"""TEST INFRASTRUCTURE ONLY -- CPU restatement of MoCo-Flow's volume-rendering path.

A from-scratch PyTorch-CPU restatement (same op sequence, own code) of the rows
of SURVEY.md §8(a):

  E  Embedding        /root/reference/models/embedding.py:4-47
  N  NeRF             /root/reference/models/nerf.py:5-102
  F  NoF              /root/reference/models/nof.py:6-85   (quaternion head through
                      oracle/kornia_restated.py -- PARITY UNPINNED for use_quat=True,
                      see that file's header)
  F' nof_inference    /root/reference/models/rendering.py:49-83
  C  nerf_inference   /root/reference/models/rendering.py:86-192
  S  sample_pdf       /root/reference/models/rendering.py:5-46
  R  render_rays      /root/reference/models/rendering.py:195-375
  +  trainer glue     /root/reference/trainer/trainer_moco_flow.py:146-187

Pinning: tests/golden/*.npz were produced by importing the reference itself in
the build container (tests/golden/gen_golden.py); tests/test_oracle_golden.py
checks this file against every one of them (<= 1e-6 relative).

It is the *checker* for the HIP path and the ``cpu_baseline`` ("port") leg of
bench.py. It is never imported by ``moco_flow_amd``.

The networks are plain containers of tensors (not nn.Module) keyed exactly like
the reference's ``state_dict`` so that the same weights load into the oracle,
the reference and the product.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from .kornia_restated import quaternion_log_to_exp, quaternion_to_rotation_matrix


def _as_tensor_dict(state) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in state.items():
        out[k] = v.detach().to(torch.float32).cpu() if isinstance(v, torch.Tensor) \
            else torch.from_numpy(v).to(torch.float32)
    return out


# --------------------------------------------------------------------------- E
class Embedding:
    """embedding.py:4-47. ``out = [x, w0*sin(f0 x), w0*cos(f0 x), w1*sin(f1 x), ...]``."""

    def __init__(self, in_channels: int, N_freqs: int, logscale: bool = True):
        self.in_channels = in_channels
        self.N_freqs = N_freqs
        self.out_channels = in_channels * (2 * N_freqs + 1)
        self.weights = [1] * N_freqs
        if logscale:
            self.freq_bands = 2 ** torch.linspace(0, N_freqs - 1, N_freqs)
        else:
            self.freq_bands = torch.linspace(1, 2 ** (N_freqs - 1), N_freqs)

    def set_weights(self, weights):
        if isinstance(weights, int):
            self.weights = [weights] * self.N_freqs
        else:
            assert len(weights) == self.N_freqs
            self.weights = weights

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        pieces = [x]
        for w, f in zip(self.weights, self.freq_bands):
            arg = f * x
            pieces.append(w * torch.sin(arg))
            pieces.append(w * torch.cos(arg))
        return torch.cat(pieces, -1)


# --------------------------------------------------------------------------- N
class NeRF:
    """nerf.py:5-102 as a tensor container."""

    def __init__(self, D=8, W=256, in_channels_xyz=33, skips=(4,), extra_feat_type="none",
                 extra_feat_dim=0, state=None):
        assert extra_feat_type in ["none", "ind", "dir", "latent_code"], \
            f"extra_feat_type {extra_feat_type} for NeRF model not supported!!!"
        self.D, self.W = D, W
        self.in_channels_xyz = in_channels_xyz
        self.skips = list(skips)
        self.extra_feat_type = extra_feat_type
        self.extra_feat_dim = extra_feat_dim
        self.p: Dict[str, torch.Tensor] = {}
        if state is not None:
            self.load_state_dict(state)

    def load_state_dict(self, state):
        self.p = _as_tensor_dict(state)

    def state_dict(self):
        return dict(self.p)

    def __call__(self, inputs, sigma_only=False, img_ind=None):
        p = self.p
        if not sigma_only:
            input_xyz, extra = torch.split(inputs, [self.in_channels_xyz, self.extra_feat_dim], dim=-1)
        else:
            input_xyz = inputs
        h = input_xyz
        for i in range(self.D):
            if i in self.skips:
                h = torch.cat([input_xyz, h], -1)
            h = F.relu(F.linear(h, p[f"xyz_encoding_{i+1}.0.weight"], p[f"xyz_encoding_{i+1}.0.bias"]))
        sigma = F.linear(h, p["sigma.weight"], p["sigma.bias"])
        if sigma_only:
            return sigma
        feat = F.linear(h, p["xyz_encoding_final.weight"], p["xyz_encoding_final.bias"])
        if self.extra_feat_type == "latent_code":
            raise NotImplementedError("NeRF model does not support latent code yet!!!")
        e = F.relu(F.linear(torch.cat([feat, extra], -1),
                            p["extra_encoding.0.weight"], p["extra_encoding.0.bias"]))
        rgb = torch.sigmoid(F.linear(e, p["rgb.0.weight"], p["rgb.0.bias"]))
        return torch.cat([rgb, sigma], -1)


# --------------------------------------------------------------------------- F
class NoF:
    """nof.py:6-85 as a tensor container."""

    def __init__(self, D=8, W=256, in_channels_xyz=33, skips=(4,), extra_feat_type="ind",
                 extra_feat_dim=0, use_quat=False, state=None):
        assert extra_feat_type in ["ind", "latent_code"], \
            f"extra_feat_type {extra_feat_type} for NoF model not supported!!!"
        self.D, self.W = D, W
        self.in_channels_xyz = in_channels_xyz
        self.skips = list(skips)
        self.extra_feat_type = extra_feat_type
        self.extra_feat_dim = extra_feat_dim
        self.use_quat = use_quat
        self.p: Dict[str, torch.Tensor] = {}
        if state is not None:
            self.load_state_dict(state)

    def load_state_dict(self, state):
        self.p = _as_tensor_dict(state)

    def state_dict(self):
        return dict(self.p)

    def __call__(self, inputs, xyz, img_ind=None):
        if self.extra_feat_type == "latent_code":
            raise NotImplementedError("NoF model does not support latent code yet!!!")
        p = self.p
        u = inputs
        for i in range(self.D):
            if i in self.skips:
                u = torch.cat([inputs, u], -1)
            u = F.relu(F.linear(u, p[f"nof_encoding_{i+1}.0.weight"], p[f"nof_encoding_{i+1}.0.bias"]))
        head = F.linear(u, p["nof_encoding_final.weight"], p["nof_encoding_final.bias"])
        if self.use_quat:
            v, s, t = head[:, :3], head[:, 3:6], head[:, 6:9]
            r = quaternion_to_rotation_matrix(quaternion_log_to_exp(v))
            return torch.bmm((xyz - s).unsqueeze(1), r).squeeze(1) + s + t
        return head + xyz


# --------------------------------------------------------------------------- S
def sample_pdf_full(bins, weights, N_importance, det=False, eps=1e-5, u=None):
    """rendering.py:5-46, returning every intermediate needed for index parity.
    ``u`` may be supplied (N, N_importance) to make the stochastic mode testable."""
    N_rays, n_w = weights.shape
    weights = weights + eps
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1)
    if u is None:
        if det:
            u = torch.linspace(0, 1, N_importance, device=bins.device).expand(N_rays, N_importance)
        else:
            u = torch.rand(N_rays, N_importance, device=bins.device)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp_min(inds - 1, 0)
    above = torch.clamp_max(inds, n_w)
    pair = torch.stack([below, above], -1).view(N_rays, 2 * N_importance)
    cdf_g = torch.gather(cdf, 1, pair).view(N_rays, N_importance, 2)
    bins_g = torch.gather(bins, 1, pair).view(N_rays, N_importance, 2)
    denom = cdf_g[..., 1] - cdf_g[..., 0]
    denom[denom < eps] = 1
    samples = bins_g[..., 0] + (u - cdf_g[..., 0]) / denom * (bins_g[..., 1] - bins_g[..., 0])
    return dict(samples=samples, cdf=cdf, u=u, inds=inds, below=below, above=above)


def sample_pdf(bins, weights, N_importance, det=False, eps=1e-5):
    return sample_pdf_full(bins, weights, N_importance, det=det, eps=eps)["samples"]


# --------------------------------------------------------------------------- F'
def _embed_padded(emb, x, width):
    e = emb(x)
    out = torch.zeros((x.shape[0], width))
    out[:, :e.shape[1]] = e
    return out


def nof_inference(xyz_, ind_, nof_embeddings, nof_model):
    """rendering.py:49-83."""
    N, S = xyz_.shape[0], xyz_.shape[1]
    flat = xyz_.reshape(-1, 3)
    xyz_e = _embed_padded(nof_embeddings[0], flat, nof_model.in_channels_xyz)
    ind_e = torch.repeat_interleave(nof_embeddings[1](ind_), repeats=S, dim=0)
    inp = torch.cat([xyz_e, ind_e], -1)
    img_ind = torch.repeat_interleave(ind_, repeats=S, dim=0).view(-1)
    return nof_model(inp, flat, img_ind=img_ind).view(N, S, -1)


# --------------------------------------------------------------------------- C
def composite(sigmas, rgbs, z_vals, dir_, noise, activate_type="relu", background=None):
    """rendering.py:157-192: sigma/rgb -> alpha, weights, rgb/depth. ``rgbs`` may be None."""
    deltas = z_vals[:, 1:] - z_vals[:, :-1]
    deltas = torch.cat([deltas, 1e10 * torch.ones_like(deltas[:, :1])], -1)
    deltas = deltas * torch.norm(dir_.unsqueeze(1), dim=-1)
    if activate_type == "relu":
        alphas = 1 - torch.exp(-deltas * torch.relu(sigmas + noise))
    elif activate_type == "softplus":
        alphas = 1 - torch.exp(-deltas * F.softplus(sigmas + noise))
    else:
        raise ValueError("activation layer type: %s not support" % activate_type)
    shifted = torch.cat([torch.ones_like(alphas[:, :1]), 1 - alphas + 1e-10], -1)
    weights = alphas * torch.cumprod(shifted, -1)[:, :-1]
    if rgbs is None:
        return weights, alphas
    wsum = weights.sum(1)
    rgb = torch.sum(weights.unsqueeze(-1) * rgbs, -2)
    depth = torch.sum(weights * z_vals, -1)
    if background is not None:
        rgb = rgb + background * (1 - wsum.unsqueeze(-1))
    return rgb, depth, weights, alphas


def nerf_inference(xyz_, ind_, dir_, z_vals, noise_std, nerf_embeddings, nerf_model,
                   background=None, weights_only=False, activate_type="relu", noise=None):
    """rendering.py:86-192. ``noise`` (N,S) overrides the internal randn draw."""
    N, S = xyz_.shape[0], xyz_.shape[1]
    dir_ = dir_.view(-1, 3)
    flat = xyz_.reshape(-1, 3)
    inp = _embed_padded(nerf_embeddings[0], flat, nerf_model.in_channels_xyz)
    if not weights_only:
        if nerf_model.extra_feat_type == "ind":
            e = torch.repeat_interleave(nerf_embeddings[1](ind_), repeats=S, dim=0)
            pad = torch.zeros((e.shape[0], nerf_model.extra_feat_dim))
            pad[:, :e.shape[1]] = e
            inp = torch.cat([inp, pad], 1)
        elif nerf_model.extra_feat_type == "dir":
            e = torch.repeat_interleave(nerf_embeddings[2](dir_), repeats=S, dim=0)
            pad = torch.zeros((e.shape[0], nerf_model.extra_feat_dim))
            pad[:, :e.shape[1]] = e
            inp = torch.cat([inp, pad], 1)
    img_ind = torch.repeat_interleave(ind_, repeats=S, dim=0).view(-1)
    out = nerf_model(inp, sigma_only=weights_only, img_ind=img_ind)
    if weights_only:
        sigmas, rgbs = out.view(N, S), None
    else:
        out = out.view(N, S, 4)
        rgbs, sigmas = out[..., :3], out[..., 3]
    if noise is None:
        noise = torch.randn(sigmas.shape) * noise_std   # always drawn (rendering.py:166)
    return composite(sigmas, rgbs, z_vals, dir_, noise, activate_type,
                     None if weights_only else background)


# --------------------------------------------------------------------------- R
def render_rays(rays, background, nerf_embeddings, nerf_models, nof_embeddings=None,
                nof_models=None, chain_local=False, chain_global=False, N_samples=64,
                N_importance=0, use_disp=False, perturb=0, noise_std=1,
                nerf_activate_type="relu", test_time=False, _capture: Optional[dict] = None,
                _z_fine_override=None, _rng: Optional[dict] = None):
    """rendering.py:195-375. ``_capture`` (test hook) receives z_vals / weights / alphas;
    ``_z_fine_override`` (test hook) replaces the sorted fine depths, so that the fine pass can be
    checked independently of the u = 1.0 resampling hazard; ``_rng`` (test hook)
    supplies the random draws (perturb_rand (N,S), noise_coarse (N,S), noise_fine (N,S+M), u (N,M))
    so that the stochastic branches (rendering.py:259, 166, 30) can be compared across devices."""
    _rng = _rng or {}
    N = rays.shape[0]
    rays_o, rays_d = rays[:, 0:3], rays[:, 3:6]
    near, far = rays[:, 6:7], rays[:, 7:8]
    img_ind = rays[:, 8:9]
    use_nof = nof_models is not None
    if use_nof and chain_global:
        chained_ind = rays[:, 9:10]

    t = torch.linspace(0, 1, N_samples)
    if not use_disp:
        z_vals = near * (1 - t) + far * t
    else:
        z_vals = 1 / (1 / near * (1 - t) + 1 / far * t)
    z_vals = z_vals.expand(N, N_samples)
    if perturb > 0:
        mid = 0.5 * (z_vals[:, :-1] + z_vals[:, 1:])
        upper = torch.cat([mid, z_vals[:, -1:]], -1)
        lower = torch.cat([z_vals[:, :1], mid], -1)
        pr = _rng["perturb_rand"] if "perturb_rand" in _rng else torch.rand(z_vals.shape)
        z_vals = lower + (upper - lower) * (perturb * pr)

    def chains(xyz):
        """bw / fw NoF evaluations of one pass (rendering.py:270-286, 335-346)."""
        bw = nof_models[0]
        canon = nof_inference(xyz, img_ind, nof_embeddings, bw)
        recon = chained_recon = None
        if chain_local and not test_time:
            fw = nof_models[1]
            recon = nof_inference(canon, img_ind, nof_embeddings, fw)
        if chain_global and not test_time:
            fw = nof_models[1] if chain_local else _unbound_fw()
            chained = nof_inference(canon, chained_ind, nof_embeddings, fw)
            chained_canon = nof_inference(chained, chained_ind, nof_embeddings, bw)
            chained_recon = nof_inference(chained_canon, img_ind, nof_embeddings, fw)
        return canon, recon, chained_recon

    def consensus(result, tag, xyz, recon, chained_recon, alphas):
        mask = alphas.ge(0.01)
        if not torch.any(mask):
            mask = torch.ones_like(mask).bool()
        if chain_local:
            result[f"nof_local_disp_{tag}"] = torch.mean(torch.abs(xyz - recon)[mask], dim=1)
        if chain_global:
            result[f"nof_global_disp_{tag}"] = torch.mean(torch.abs(xyz - chained_recon)[mask], dim=1)

    xyz_c = rays_o.unsqueeze(1) + rays_d.unsqueeze(1) * z_vals.unsqueeze(2)
    if use_nof:
        nerf_in, recon_c, chained_recon_c = chains(xyz_c)
    else:
        nerf_in = xyz_c

    if N_importance > 0 and test_time:
        w_c, a_c = nerf_inference(nerf_in, img_ind, rays_d, z_vals, noise_std, nerf_embeddings,
                                  nerf_models[0], background=background, weights_only=True,
                                  activate_type=nerf_activate_type, noise=_rng.get("noise_coarse"))
        result = {"opacity_coarse": w_c.sum(1)}
    else:
        rgb_c, depth_c, w_c, a_c = nerf_inference(nerf_in, img_ind, rays_d, z_vals, noise_std,
                                                  nerf_embeddings, nerf_models[0],
                                                  background=background, weights_only=False,
                                                  activate_type=nerf_activate_type,
                                                  noise=_rng.get("noise_coarse"))
        result = {"rgb_coarse": rgb_c, "depth_coarse": depth_c, "opacity_coarse": w_c.sum(1)}
    if _capture is not None:
        _capture.update(z_coarse=z_vals, weights_coarse=w_c, alphas_coarse=a_c)

    if use_nof and not test_time:
        consensus(result, "coarse", xyz_c, recon_c, chained_recon_c, a_c)

    if N_importance > 0:
        mid = 0.5 * (z_vals[:, :-1] + z_vals[:, 1:])
        if "u" in _rng:
            z_new = sample_pdf_full(mid, w_c[:, 1:-1], N_importance, det=False, u=_rng["u"])["samples"].detach()
        else:
            z_new = sample_pdf(mid, w_c[:, 1:-1], N_importance, det=(perturb == 0)).detach()
        z_vals, _ = torch.sort(torch.cat([z_vals, z_new], -1), -1)
        if _z_fine_override is not None:
            z_vals = _z_fine_override
        xyz_f = rays_o.unsqueeze(1) + rays_d.unsqueeze(1) * z_vals.unsqueeze(2)
        if use_nof:
            nerf_in, recon_f, chained_recon_f = chains(xyz_f)
        else:
            nerf_in = xyz_f
        rgb_f, depth_f, w_f, a_f = nerf_inference(nerf_in, img_ind, rays_d, z_vals, noise_std,
                                                  nerf_embeddings, nerf_models[1],
                                                  background=background, weights_only=False,
                                                  activate_type=nerf_activate_type, noise=_rng.get("noise_fine"))
        result["rgb_fine"] = rgb_f
        result["depth_fine"] = depth_f
        result["opacity_fine"] = w_f.sum(1)
        if _capture is not None:
            _capture.update(z_fine=z_vals, weights_fine=w_f, alphas_fine=a_f)
        if use_nof and not test_time:
            consensus(result, "fine", xyz_f, recon_f, chained_recon_f, a_f)
    return result


def _unbound_fw():
    # rendering.py:276-280: fw_nof is bound only under chain_local; chain_global alone
    # hits an unbound local in the reference [SURVEY.md §8a row R].
    raise UnboundLocalError("local variable 'fw_nof' referenced before assignment")


# ------------------------------------------------------------- trainer-side glue
def forward_nerf_alpha(xyz, deltas, nerf_embedding_xyz, nerf_model):
    """trainer_moco_flow.py:146-157 (``forwarf_nerf``): softplus-activated alpha of points."""
    inp = _embed_padded(nerf_embedding_xyz, xyz, nerf_model.in_channels_xyz)
    sigmas = nerf_model(inp, sigma_only=True)
    return 1 - torch.exp(-deltas * F.softplus(sigmas))


def forward_nof_points(xyz, ind, num_frames, nof_embedding_xyz, nof_embedding_ind, nof_model):
    """trainer_moco_flow.py:159-187 (``forward_nof``): ind is a frame index tensor (1,)."""
    inp = _embed_padded(nof_embedding_xyz, xyz, nof_model.in_channels_xyz)
    if nof_model.extra_feat_type == "ind":
        ind_f = ind.unsqueeze(0).repeat((xyz.shape[0], 1)).float() * 2 / num_frames - 1.0
        e = nof_embedding_ind(ind_f)
        pad = torch.zeros((xyz.shape[0], nof_model.extra_feat_dim))
        pad[:, :e.shape[1]] = e
        inp = torch.cat([inp, pad], -1)
    return nof_model(inp, xyz, ind)


def psnr(a, b):
    """models/metrics.py:4-13."""
    return -10 * torch.log10(torch.mean((a - b) ** 2))


# ------------------------------------------------------------------ conveniences
def build_nerf(state, D=8, W=256, in_channels_xyz=63, skips=(4,), extra_feat_type="dir",
               extra_feat_dim=27):
    return NeRF(D, W, in_channels_xyz, skips, extra_feat_type, extra_feat_dim, state=state)


def build_nof(state, D=4, W=128, in_channels_xyz=33, skips=(2,), extra_feat_dim=33, use_quat=True):
    return NoF(D, W, in_channels_xyz, skips, "ind", extra_feat_dim, use_quat, state=state)


# ------------------------------------------------------------- producers (§8f rows 3-4)
def gen_ray_directions(H, W, focal, camera_c=(0, 0)):
    """utils/camera.py:29-50."""
    i, j = torch.meshgrid(torch.linspace(0, W - 1, W), torch.linspace(0, H - 1, H), indexing="ij")
    i, j = i.t(), j.t()
    focal = [focal[0], focal[0]] if len(focal) == 1 else list(focal)
    return torch.stack([(i - camera_c[0]) / focal[0], -(j - camera_c[1]) / focal[0], -torch.ones_like(i)], -1)


def make_rays(H, W, focal, center, c2w, near, far, idx):
    """utils/camera.py:52-81 + 134-148: (H*W, 9) rays. c2w: (3|4, 4) numpy array or None."""
    directions = gen_ray_directions(H, W, focal, center)
    if c2w is None:
        rays_d = directions / torch.norm(directions, dim=-1, keepdim=True)
        rays_o = torch.zeros_like(directions)
    else:
        m = torch.from_numpy(np.asarray(c2w)[:3, :4]).float()
        rays_d = directions @ m[:, :3].T
        rays_d = rays_d / torch.norm(rays_d, dim=-1, keepdim=True)
        rays_o = m[:, 3].expand(rays_d.shape)
    rays_d, rays_o = rays_d.reshape(-1, 3), rays_o.reshape(-1, 3)
    one = torch.ones_like(rays_o[:, :1])
    return torch.cat([rays_o, rays_d, near * one, far * one, idx * one], dim=1)


def project_aabb(aabb_verts, c2w, K):
    """utils/camera.py:83-103 (calculate_2d_projections): world points -> integer pixel coordinates (x = column,
    y = row); the cast to int32 truncates toward zero."""
    pts = np.asarray(aabb_verts).transpose()
    homo = np.vstack([pts, np.ones((1, pts.shape[1]), dtype=np.float32)])
    cam = np.linalg.inv(np.asarray(c2w)) @ homo
    cam = cam[:3, :] / cam[3, :]
    cam[1:, :] *= -1
    pix = np.asarray(K) @ cam[:3, :]
    pix = (pix[:2, :] / pix[2, :]).transpose()
    return np.array(pix, dtype=np.int32)


def convex_hull_int(points):
    """Convex hull of integer points, counter-clockwise in (x, y) with y down-screen irrelevant: Andrew's monotone
    chain, collinear points dropped.  (cv2.convexHull, utils/camera.py:123, returns the same vertex set.)"""
    pts = sorted(set((int(x), int(y)) for x, y in points))
    if len(pts) <= 2:
        return pts

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])

    lower, upper = [], []
    for q in pts:
        while len(lower) >= 2 and cross(lower[-2], lower[-1], q) <= 0:
            lower.pop()
        lower.append(q)
    for q in reversed(pts):
        while len(upper) >= 2 and cross(upper[-2], upper[-1], q) <= 0:
            upper.pop()
        upper.append(q)
    return lower[:-1] + upper[:-1]


def _cv_clip_line(W, H, x1, y1, x2, y2):
    """cv::clipLine (OpenCV modules/imgproc/src/drawing.cpp) on [0, W-1] x [0, H-1]; None when nothing is inside."""
    right, bottom = W - 1, H - 1
    if W <= 0 or H <= 0:
        return None
    code = lambda x, y: (x < 0) + (x > right) * 2 + (y < 0) * 4 + (y > bottom) * 8
    c1, c2 = code(x1, y1), code(x2, y2)
    if (c1 & c2) == 0 and (c1 | c2) != 0:
        if c1 & 12:
            a = 0 if c1 < 8 else bottom
            x1 += int(float(a - y1) * (x2 - x1) / (y2 - y1))       # (int64)((double)(a - y1) * (x2 - x1) / (y2 - y1))
            y1 = a
            c1 = (x1 < 0) + (x1 > right) * 2
        if c2 & 12:
            a = 0 if c2 < 8 else bottom
            x2 += int(float(a - y2) * (x2 - x1) / (y2 - y1))
            y2 = a
            c2 = (x2 < 0) + (x2 > right) * 2
        if (c1 & c2) == 0 and (c1 | c2) != 0:
            if c1:
                a = 0 if c1 == 1 else right
                y1 += int(float(a - x1) * (y2 - y1) / (x2 - x1))
                x1 = a
                c1 = 0
            if c2:
                a = 0 if c2 == 1 else right
                y2 += int(float(a - x2) * (y2 - y1) / (x2 - x1))
                x2 = a
                c2 = 0
    return (x1, y1, x2, y2) if (c1 | c2) == 0 else None


def _cv_line(mask, p1, p2):
    """cv::Line(img, p1, p2, color, 8): clipLine, then LineIterator(connectivity 8, leftToRight = true) -- the
    error-term walk as OpenCV runs it, one pixel per step."""
    H, W = mask.shape
    c = _cv_clip_line(W, H, p1[0], p1[1], p2[0], p2[1])
    if c is None:
        return
    x1, y1, x2, y2 = c
    dx, dy = x2 - x1, y2 - y1
    if dx < 0:                       # leftToRight: start from the leftmost endpoint
        dx, dy, x1, y1 = -dx, -dy, x2, y2
    step_x, step_y = 1, 1
    if dy < 0:
        dy, step_y = -dy, -1
    vert = dy > dx
    if vert:
        dx, dy = dy, dx
    err, plus_delta, minus_delta = dx - (dy + dy), dx + dx, -(dy + dy)
    x, y = x1, y1
    for _ in range(dx + 1):
        mask[y, x] = True
        diag = err < 0
        err += minus_delta + (plus_delta if diag else 0)
        if vert:
            y += step_y
            x += step_x if diag else 0
        else:
            x += step_x
            y += step_y if diag else 0


def valid_rays_mask(projected_pixels, H, W):
    """Camera.get_valid_rays_mask (utils/camera.py:119-132): cv2.fillConvexPoly(zeros, cv2.convexHull(pts), 255) > 0.
    PARITY UNPINNED vs cv2 (absent from this image).  This restates OpenCV's FillConvexPoly (modules/imgproc/src/
    drawing.cpp; line_type 8, shift 0) the way OpenCV runs it -- sequential loops -- where the kernel uses closed forms:
    first every hull edge is drawn with Line() (clipLine + the 8-connected LineIterator), then the scan-line loop fills
    hlines between two edge chains stepped in 16.16 fixed point with rounded slopes, and stops when it runs out of edges
    -- i.e. BEFORE the hull's last row, which only the outline covers."""
    hull = convex_hull_int(projected_pixels)
    mask = np.zeros((H, W), dtype=bool)
    n = len(hull)
    if n == 0:
        return mask.reshape(-1)
    XY_SHIFT, XY_ONE = 16, 1 << 16
    v = [(int(x), int(y)) for x, y in hull]
    ymin, ymax, imin = v[0][1], v[0][1], 0
    p0 = v[n - 1]
    for i, p in enumerate(v):
        if p[1] < ymin:
            ymin, imin = p[1], i
        ymax = max(ymax, p[1])
        _cv_line(mask, p0, p)
        p0 = p
    xs_all = [q[0] for q in v]
    if n < 3 or max(xs_all) < 0 or ymax < 0 or min(xs_all) >= W or ymin >= H:
        return mask.reshape(-1)
    ymax = min(ymax, H - 1)
    edge = [dict(idx=imin, ye=ymin, di=1, x=-XY_ONE, dx=0), dict(idx=imin, ye=ymin, di=n - 1, x=-XY_ONE, dx=0)]
    edges, y = n, ymin

    def trunc_div(a, b):             # C integer division (truncates toward zero)
        q = abs(a) // abs(b)
        return q if (a < 0) == (b < 0) else -q

    while True:
        for e in edge:
            if y >= e["ye"]:
                idx0, di = e["idx"], e["di"]
                idx = idx0 + di
                if idx >= n:
                    idx -= n
                while True:
                    edges -= 1
                    if edges < 0:
                        break
                    ty = v[idx][1]
                    if ty > y:
                        xs, xe = v[idx0][0] << XY_SHIFT, v[idx][0] << XY_SHIFT
                        e["ye"] = ty
                        e["dx"] = trunc_div((xe - xs) * 2 + (ty - y), 2 * (ty - y))
                        e["x"] = xs
                        e["idx"] = idx
                        break
                    idx0 = idx
                    idx += di
                    if idx >= n:
                        idx -= n
        if edges < 0:
            break
        if y >= 0:
            left, right = (0, 1) if edge[0]["x"] <= edge[1]["x"] else (1, 0)
            xx1 = (edge[left]["x"] + (XY_ONE >> 1)) >> XY_SHIFT
            xx2 = (edge[right]["x"] + (XY_ONE >> 1)) >> XY_SHIFT
            if xx2 >= 0 and xx1 < W:
                mask[y, max(xx1, 0):min(xx2, W - 1) + 1] = True
        edge[0]["x"] += edge[0]["dx"]
        edge[1]["x"] += edge[1]["dx"]
        y += 1
        if y > ymax:
            break
    return mask.reshape(-1)


def knn1(ref, query):
    """k = 1 brute force with the wheel's semantics (knn_cuda/csrc/cuda/knn.cu:29-183, __init__.py:42-46):
    Euclidean distance and 0-based index of the nearest reference point, FIRST minimum on ties.
    PARITY UNPINNED against the wheel itself (CUDA only, cannot run here); exact by construction."""
    d2 = ((query[:, None, :].double() - ref[None, :, :].double()) ** 2).sum(-1)
    ind = torch.argmin(d2, dim=1)              # first minimum
    return torch.sqrt(d2.gather(1, ind[:, None])).float(), ind[:, None]


def render_image(rays, background, render, N_rand, rays_msk=None):
    """MoCoFlowTrainer.render, trainer/trainer_moco_flow.py:226-268 (= NeRFTrainer.render,
    trainer_nerf.py:100-140), with ``render(rays_chunk, background_chunk)`` standing for ``self.forward``:
    numpy mask selection, the chunk loop over the UNMASKED ray count (empty trailing chunks included),
    and the foreground scatter-back.  ``np.float`` of :258 (removed from NumPy >= 1.24) is float64."""
    import numpy as np
    from collections import defaultdict
    if rays_msk is not None:
        msk = np.where(rays_msk == True)  # noqa: E712  (as in the reference)
        rendered_rays, rendered_background = rays[msk], background[msk]
    else:
        rendered_rays, rendered_background = rays, background
    B = rays.shape[0]
    results = defaultdict(list)
    for i in range(0, B, N_rand):
        chunk = render(rendered_rays[i:i + N_rand], rendered_background[i:i + N_rand])
        for k, v in chunk.items():
            results[k] += [v]
    results = {k: torch.cat(v, 0) for k, v in results.items()}
    if rays_msk is not None:
        typ = "fine" if "rgb_fine" in results else "coarse"
        num_ori_rays = rays.shape[0]
        img_raw = torch.zeros(num_ori_rays, 3)
        depth_raw = torch.ones(num_ori_rays) * 10
        opacity = results["opacity_%s" % typ].cpu().numpy()
        foreground_idx = np.where(opacity > 0)
        foreground_mask = np.zeros_like(rays_msk).astype(np.float64)
        foreground_mask[msk] = opacity
        img_raw[foreground_mask > 0] = results["rgb_%s" % typ][foreground_idx]
        depth_raw[msk] = 8
        depth_raw[foreground_mask > 0] = results["depth_%s" % typ][foreground_idx]
        img_raw[foreground_mask == 0] = background[foreground_mask == 0]
        results["rgb_%s" % typ] = img_raw
        results["depth_%s" % typ] = depth_raw
    return results
