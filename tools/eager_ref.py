"""Eager PyTorch-ROCm restatement of the reference's op sequence on the drop-in's own modules -- TOOLS ONLY (timing
comparisons: "what the reference itself would run on this GPU"; tools/time_*.py).  Not imported by the package, not a
fallback, not a checker (the checker is oracle/cpu_ref.py on the CPU).  References: models/embedding.py:42-46,
models/nerf.py:78-102, models/nof.py:69-82, models/rendering.py:49-192, 245-375."""
from typing import Dict, Optional

import torch
import torch.nn.functional as F

def embed(emb, x):
    """embedding.py:42-46 with torch ops on x's device."""
    out = [x]
    for w, f in zip(emb.weights, emb.freq_bands):
        f = float(f)
        out += [w * torch.sin(f * x), w * torch.cos(f * x)]
    return torch.cat(out, -1)


def _pad_to(t, width):
    if t.shape[1] == width:
        return t
    out = t.new_zeros((t.shape[0], width))
    out[:, :t.shape[1]] = t
    return out


def nerf_forward(m, inputs, sigma_only=False):
    """nerf.py:78-102 on the module's own parameters."""
    if not sigma_only:
        xyz, extra = torch.split(inputs, [m.in_channels_xyz, m.extra_feat_dim], dim=-1)
    else:
        xyz = inputs
    h = xyz
    for i in range(m.D):
        if i in m.skips:
            h = torch.cat([xyz, h], -1)
        lin = getattr(m, f"xyz_encoding_{i+1}")[0]
        h = F.relu(F.linear(h, lin.weight, lin.bias))
    sigma = F.linear(h, m.sigma.weight, m.sigma.bias)
    if sigma_only:
        return sigma
    feat = F.linear(h, m.xyz_encoding_final.weight, m.xyz_encoding_final.bias)
    e = F.relu(F.linear(torch.cat([feat, extra], -1), m.extra_encoding[0].weight, m.extra_encoding[0].bias))
    rgb = torch.sigmoid(F.linear(e, m.rgb[0].weight, m.rgb[0].bias))
    return torch.cat([rgb, sigma], -1)


def _quat_rotate(T, xyz):
    """kornia 0.6.5 quaternion_log_to_exp + quaternion_to_rotation_matrix (restated),
    then nof.py:80."""
    v, s, t = T[:, :3], T[:, 3:6], T[:, 6:9]
    n = torch.norm(v, p=2, dim=-1, keepdim=True).clamp(min=1e-8)
    q = torch.cat([v * torch.sin(n) / n, torch.cos(n)], -1)
    q = F.normalize(q, p=2.0, dim=-1, eps=1e-12)
    x, y, z, w = torch.chunk(q, 4, dim=-1)
    tx, ty, tz = 2.0 * x, 2.0 * y, 2.0 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    R = torch.stack((1.0 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1.0 - (txx + tzz), tyz - twx,
                     txz - twy, tyz + twx, 1.0 - (txx + tyy)), dim=-1).view(-1, 3, 3)
    return torch.bmm((xyz - s).unsqueeze(1), R).squeeze(1) + s + t


def nof_forward(m, inputs, xyz):
    """nof.py:69-82."""
    u = inputs
    for i in range(m.D):
        if i in m.skips:
            u = torch.cat([inputs, u], -1)
        lin = getattr(m, f"nof_encoding_{i+1}")[0]
        u = F.relu(F.linear(u, lin.weight, lin.bias))
    head = F.linear(u, m.nof_encoding_final.weight, m.nof_encoding_final.bias)
    return _quat_rotate(head, xyz) if m.use_quat else head + xyz


def _nof_points(xyz, ind, nof_embs, m):
    """rendering.py:49-83 for (N,S,3) points and (N,1) indices."""
    N, S = xyz.shape[0], xyz.shape[1]
    flat = xyz.reshape(-1, 3)
    xe = _pad_to(embed(nof_embs[0], flat), m.in_channels_xyz)
    ie = torch.repeat_interleave(embed(nof_embs[1], ind), repeats=S, dim=0)
    return nof_forward(m, torch.cat([xe, ie], -1), flat).view(N, S, 3)


def render_pass(rays, background, z_vals, noise, activation, nerf, nerf_embs, nof_models, nof_embs,
                chain_local, chain_global, sigma_only, masks: Optional[Dict[str, torch.Tensor]]):
    """One pass of render_rays (rendering.py:262-314 / 329-373) as differentiable torch ops on given
    depths. Returns the same dict the fused kernel fills (rgb, depth, opacity, weights, alphas and the
    mask-compacted consensus vectors when ``masks`` carries the kernel's mask)."""
    N, S = z_vals.shape
    o, d = rays[:, 0:3], rays[:, 3:6]
    ind = rays[:, 8:9]
    xyz = o.unsqueeze(1) + d.unsqueeze(1) * z_vals.unsqueeze(2)
    out = {}
    pts = xyz
    if nof_models is not None:
        bw = nof_models[0]
        canon = _nof_points(xyz, ind, nof_embs, bw)
        if chain_local:
            fw = nof_models[1]
            recon = _nof_points(canon, ind, nof_embs, fw)
            out["disp_local_full"] = torch.abs(xyz - recon)
        if chain_global:
            cind = rays[:, 9:10]
            a = _nof_points(canon, cind, nof_embs, fw)
            b = _nof_points(a, cind, nof_embs, bw)
            out["disp_global_full"] = torch.abs(xyz - _nof_points(b, ind, nof_embs, fw))
        pts = canon
    flat = pts.reshape(-1, 3)
    inp = _pad_to(embed(nerf_embs[0], flat), nerf.in_channels_xyz)
    if not sigma_only:
        if nerf.extra_feat_type == "ind":
            e = torch.repeat_interleave(embed(nerf_embs[1], ind), repeats=S, dim=0)
            inp = torch.cat([inp, _pad_to(e, nerf.extra_feat_dim)], 1)
        elif nerf.extra_feat_type == "dir":
            e = torch.repeat_interleave(embed(nerf_embs[2], d), repeats=S, dim=0)
            inp = torch.cat([inp, _pad_to(e, nerf.extra_feat_dim)], 1)
    net = nerf_forward(nerf, inp, sigma_only=sigma_only)
    if sigma_only:
        sigmas, rgbs = net.view(N, S), None
    else:
        net = net.view(N, S, 4)
        rgbs, sigmas = net[..., :3], net[..., 3]
    deltas = z_vals[:, 1:] - z_vals[:, :-1]
    deltas = torch.cat([deltas, 1e10 * torch.ones_like(deltas[:, :1])], -1) * torch.norm(d.unsqueeze(1), dim=-1)
    sg = sigmas if noise is None else sigmas + noise
    act = torch.relu(sg) if activation == "relu" else F.softplus(sg)
    alphas = 1 - torch.exp(-deltas * act)
    shifted = torch.cat([torch.ones_like(alphas[:, :1]), 1 - alphas + 1e-10], -1)
    weights = alphas * torch.cumprod(shifted, -1)[:, :-1]
    out["opacity"] = weights.sum(1)
    out["weights"], out["alphas"] = weights, alphas
    if not sigma_only:
        rgb = torch.sum(weights.unsqueeze(-1) * rgbs, -2)
        if background is not None:
            rgb = rgb + background * (1 - out["opacity"].unsqueeze(-1))
        out["rgb"] = rgb
        out["depth"] = torch.sum(weights * z_vals, -1)
    return out



def composite_from_samples(rgbsig, z_vals, rays_d, noise, activation, background, sigma_only):
    """rendering.py:157-192 on per-sample (rgb, sigma) planes -- differentiable, (N,S) elementwise only."""
    N, S = z_vals.shape
    rs = rgbsig.view(N, S, 4)
    rgbs, sigmas = rs[..., :3], rs[..., 3]
    deltas = z_vals[:, 1:] - z_vals[:, :-1]
    deltas = torch.cat([deltas, 1e10 * torch.ones_like(deltas[:, :1])], -1) * torch.norm(rays_d.unsqueeze(1), dim=-1)
    sg = sigmas if noise is None else sigmas + noise
    act = torch.relu(sg) if activation == "relu" else F.softplus(sg)
    alphas = 1 - torch.exp(-deltas * act)
    shifted = torch.cat([torch.ones_like(alphas[:, :1]), 1 - alphas + 1e-10], -1)
    weights = alphas * torch.cumprod(shifted, -1)[:, :-1]
    out = {"opacity": weights.sum(1), "alphas": alphas, "weights": weights}
    if not sigma_only:
        rgb = torch.sum(weights.unsqueeze(-1) * rgbs, -2)
        if background is not None:
            rgb = rgb + background * (1 - out["opacity"].unsqueeze(-1))
        out["rgb"] = rgb
        out["depth"] = torch.sum(weights * z_vals, -1)
    return out


def render_rays_eager(rays, background, nerf_embs, nerf_models, nof_embs=None, nof_models=None, chain_local=False,
                      chain_global=False, N_samples=64, N_importance=0, perturb=0, nerf_activate_type="relu"):
    """render_rays (rendering.py:195-375, training mode, noise_std = 0) with differentiable device ops; only the detached
    hierarchical resample goes through the package (moco_flow_amd.resample_merge)."""
    from moco_flow_amd import resample_merge
    dev, N, S = rays.device, rays.shape[0], N_samples
    near, far = rays[:, 6:7], rays[:, 7:8]
    t = torch.linspace(0, 1, S, device=dev)
    z = near * (1 - t) + far * t
    if perturb > 0:
        mid = 0.5 * (z[:, :-1] + z[:, 1:])
        upper, lower = torch.cat([mid, z[:, -1:]], -1), torch.cat([z[:, :1], mid], -1)
        z = lower + (upper - lower) * (perturb * torch.rand(z.shape, device=dev))
    result = {}

    def mask_of(alphas):
        mask = alphas.ge(0.01)
        return mask if bool(torch.any(mask)) else torch.ones_like(mask)

    def one(tag, nerf, zz):
        r = render_pass(rays, background, zz, None, nerf_activate_type, nerf, nerf_embs, nof_models, nof_embs, chain_local,
                        chain_global, False, None)
        result[f"rgb_{tag}"], result[f"depth_{tag}"], result[f"opacity_{tag}"] = r["rgb"], r["depth"], r["opacity"]
        if chain_local or chain_global:
            mask = mask_of(r["alphas"].detach())
            if chain_local:
                result[f"nof_local_disp_{tag}"] = torch.mean(r["disp_local_full"][mask], dim=1)
            if chain_global:
                result[f"nof_global_disp_{tag}"] = torch.mean(r["disp_global_full"][mask], dim=1)
        return r

    c = one("coarse", nerf_models[0], z)
    if N_importance > 0:
        z_all = resample_merge(z.contiguous(), c["weights"].detach(), N_importance, det=(perturb == 0))
        one("fine", nerf_models[1], z_all)
    return result
