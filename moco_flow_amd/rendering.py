"""Drop-in for /root/reference/models/rendering.py: ``render_rays`` (:195-375) and
``sample_pdf`` (:5-46) with the reference's signatures, running on the HIP path.

Each pass (coarse, fine) is ONE fused kernel launch (mf_render_pass): ray -> depths ->
points -> [NoF chains] -> encoding -> NeRF -> composite (bf16 modes with NoF: + one small
launch per render_rays call for the per-ray index bias, mf_render_prepare). The hierarchical
resample is one launch (mf_sample_pdf), the consensus vectors are lazy (lazy.MaskedVector: their
means come from the two launches of mf_loss_partials; the compaction, mf_compact_mask, runs only
when a vector's length is asked for). Random draws (perturb / noise / stochastic resampling) are made
here with torch, in the reference's order, and handed to the kernels, which stay deterministic.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from . import autograd as A

# Arithmetic of the hidden (256-/128-wide) GEMMs of the fused pass: "f32" = exact-fp32 MFMA (the
# reference's arithmetic, BASELINE configs C1-C2; default); "bf16" = bf16 operands with fp32
# accumulate (BASELINE configs C3-C5), the NoF's xyz block as a two-term bf16 split (16 mantissa
# bits), its image-index block as an exact fp32 per-ray bias, heads and composite in fp32;
# "bf16x3" = the fp32 contract on the bf16 pipe (include/mocoflow_hip.h, MF_PREC_BF16X3: the NeRF's matrix products as three
# bf16 products of (hi, lo) operand pairs, the NoFs' as three of IEEE-half (hi, lo) pairs, fp32 accumulation, heads on the fp32
# accumulators: 1e-4 max-rel on every per-ray output -- <= 5e-5 on the golden vectors, <= 3.1e-5 through the MoCo chains at
# 4096 rays -- at 0.35-0.4 of the fp32 kernels' time).
# The reference's render_rays signature has no such knob, so it is a module setting.
PRECISION = "f32"
# Arithmetic of the TRAINING forward (a pass that records gradients; round 5: passes with NoF too):
# "f32" = the reference's; "bf16x3" = the three-product kernels writing the same activation dump (forward
# values and dumped activations to ~1e-5; the dX chain on them stays fp32, the weight gradients follow set_wgrad_precision).
TRAIN_FORWARD_PRECISION = "f32"


def set_train_forward_precision(p: str):
    global TRAIN_FORWARD_PRECISION
    if p not in ("f32", "bf16x3"):
        raise ValueError(f"training-forward precision must be 'f32' or 'bf16x3', got {p!r}")
    TRAIN_FORWARD_PRECISION = p


def set_precision(p: str):
    global PRECISION
    if p not in L.PRECISIONS:
        raise ValueError(f"precision must be one of {sorted(L.PRECISIONS)}, got {p!r}")
    PRECISION = p


# Training (gradients requested): forward values come from the fused HIP kernels, which also dump what the backward needs;
# the backward is hand-written HIP (autograd.py): mf_loss_partials_backward, mf_composite_backward, mf_nerf_backward_x
# (+ mf_embedding_backward under NoF), mf_nof_backward, mf_weight_grads.  A pass that records gradients runs the fp32
# kernels whatever the module setting (the reference trains in fp32; bf16 / bf16x3 are modes of gradient-free passes).
# What the HIP backward is not built for raises NotImplementedError under grad -- network shapes outside the envelope
# (autograd.require_nerf_hip / require_nof_hip) and test_time passes (their coarse pass evaluates sigma only; the
# reference renders those under no_grad).  There is no eager fallback in the package.


# The consensus vectors nof_*_disp_* as lazy.MaskedVector (default): .mean() / .sum() from masked sums on the device, the
# data-dependent-length tensor only when something asks for it.  False: eager tensors (compaction + host sync per pass).
LAZY_CONSENSUS = True
# torch.mean of a training pass's consensus vector as one autograd node on the distances the fused pass wrote
# (autograd.ConsensusMean); False: torch ops on the per-sample planes (|x - recon|, mask, masked sums), as in rounds 2-3
FUSED_CONSENSUS_MEAN = True


# Draw torch.randn(N,S) in every pass even when noise_std == 0, as the reference does
# (rendering.py:166), so that the device RNG stream advances identically.
STRICT_RNG = True


_LINSPACE = {}


def _linspace01(S, dev):
    """torch.linspace(0, 1, S) on the device, built once per (S, device): a launch per render_rays call otherwise."""
    key = (S, str(dev))
    t = _LINSPACE.get(key)
    if t is None:
        t = _LINSPACE[key] = torch.linspace(0, 1, S, device=dev)
    return t


def _emb_desc(e):
    return e.descriptor() if e is not None else L.mf_embedding()


def _render_pass(rays, background, z_vals, z_steps, use_disp, noise, activation, nerf, nerf_embs,
                 nof_models, nof_embs, chain_local, chain_global, sigma_only, want_planes, dump=False, precision=None,
                 workspace=None):
    """One mf_render_pass call. Returns dict of fresh tensors.  ``precision``: None = the module setting.
    ``workspace``: a one-element list shared by the passes of one render_rays call -- the first pass that needs the bf16
    NoF's per-ray bias table allocates and fills it (mf_render_prepare), the next pass (same rays, same NoFs, same chain
    flags) reuses it; None: prepare for this pass alone."""
    dev = rays.device
    N = rays.shape[0]
    S = z_vals.shape[1] if z_vals is not None else z_steps.shape[0]
    # the reference zero-pads each embedding INTO the network's input width (rendering.py:70-72, 127-142) and fails with a
    # shape error when the embedding is wider; the NoF's index block is concatenated unpadded (:73-75) and must fit exactly
    extra_emb = nerf_embs[1] if nerf.extra_feat_type == "ind" else (nerf_embs[2] if nerf.extra_feat_type == "dir" else None)
    if nerf_embs[0].out_channels > nerf.in_channels_xyz or \
            (extra_emb is not None and not sigma_only and extra_emb.out_channels > nerf.extra_feat_dim):
        raise RuntimeError(f"render_rays: embedding wider than the NeRF's input block (xyz {nerf_embs[0].out_channels} > "
                           f"{nerf.in_channels_xyz} or {nerf.extra_feat_type} > {nerf.extra_feat_dim})")
    if nof_models is not None:
        for m in nof_models[:2 if (chain_local or chain_global) else 1]:
            if nof_embs[0].out_channels > m.in_channels_xyz or nof_embs[1].out_channels != m.extra_feat_dim:
                raise RuntimeError(f"render_rays: the NoF takes [xyz <= {m.in_channels_xyz} | ind = {m.extra_feat_dim}] input columns, "
                                   f"the embeddings give {nof_embs[0].out_channels} | {nof_embs[1].out_channels}")
    a = L.mf_render_args()
    a.rays, a.ray_stride, a.n_rays = L.ptr(rays), rays.stride(0), N
    a.background = L.ptr(background)
    a.n_samples = S
    a.z_vals, a.z_steps, a.use_disp = L.ptr(z_vals), L.ptr(z_steps), 1 if use_disp else 0
    a.noise = L.ptr(noise)
    a.activation = activation
    flags = 0
    if sigma_only:
        flags |= L.MF_F_SIGMA_ONLY
    if chain_local:
        flags |= L.MF_F_CHAIN_LOCAL
    if chain_global:
        flags |= L.MF_F_CHAIN_GLOBAL
    a.flags = flags
    prec = L.PRECISIONS[precision or PRECISION]
    a.precision = prec
    desc, buf = nerf.packed(prec)
    a.nerf, a.nerf_packed = C.pointer(desc), buf.data_ptr()
    a.emb_xyz = _emb_desc(nerf_embs[0])
    if nerf.extra_feat_type == "ind":
        a.emb_extra = _emb_desc(nerf_embs[1])
    elif nerf.extra_feat_type == "dir":
        a.emb_extra = _emb_desc(nerf_embs[2])
    keep = [desc, buf]
    if nof_models is not None:
        bd, bb = nof_models[0].packed(prec)
        a.nof_bw, a.nof_bw_packed = C.pointer(bd), bb.data_ptr()
        keep += [bd, bb]
        if chain_local or chain_global:
            fd, fb = nof_models[1].packed(prec)
            a.nof_fw, a.nof_fw_packed = C.pointer(fd), fb.data_ptr()
            keep += [fd, fb]
        a.nof_emb_xyz, a.nof_emb_ind = _emb_desc(nof_embs[0]), _emb_desc(nof_embs[1])
    out = {}

    def alloc(name, shape, cond=True):
        if cond:
            out[name] = torch.empty(shape, device=dev, dtype=torch.float32)
            return out[name].data_ptr()
        return None

    a.rgb = alloc("rgb", (N, 3), not sigma_only)
    a.depth = alloc("depth", (N,), not sigma_only)
    a.opacity = alloc("opacity", (N,))
    a.weights = alloc("weights", (N, S), want_planes)
    a.alphas = alloc("alphas", (N, S), want_planes)
    a.disp_local = alloc("disp_local", (N, S), chain_local)
    a.disp_global = alloc("disp_global", (N, S), chain_global)
    if dump:                                     # training forward: what the explicit backward reads
        stride = (nerf.D + 1) * nerf.W + nerf.W // 2
        a.dump_acts = alloc("acts", (N * S, stride), not sigma_only)
        a.dump_stride = stride
        a.dump_rgbsigma = alloc("rgbsig", (N * S, 4))
        a.dump_xyz = alloc("xyz_in", (N * S, 3))
        if not sigma_only and prec in (L.MF_PREC_F32, L.MF_PREC_BF16X3) and A.DX_PRECISION == "bf16x3":
            # the ReLU bit mask of the dumped activations: all the three-product dX chain needs of them (32 bytes instead
            # of 1 KiB per layer and sample); travels with the dump tensor
            mw = (nerf.D + 2) * 8
            out["acts"]._mf_mask = torch.empty((N * S, mw), device=dev, dtype=torch.int32)
            a.dump_mask, a.dump_mask_stride = out["acts"]._mf_mask.data_ptr(), mw
        if nof_models is not None and len({(m.D, m.W) for m in nof_models}) == 1:
            # the chain's NoF evaluations dump too (one plane per step): no re-evaluation in the backward graph
            steps = 1 + (1 if chain_local else 0) + (3 if chain_global else 0)
            # planes grouped by network (bw: steps 0, 3; fw: steps 1, 2, 4): one weight-gradient contraction per network
            order = [k for k in range(steps) if k in (0, 3)] + [k for k in range(steps) if k in (1, 2, 4)]
            out["nof_plane"] = [order.index(k) for k in range(steps)]
            for k in range(steps):
                a.dump_nof_plane[k] = out["nof_plane"][k]
            x3 = prec == L.MF_PREC_BF16X3
            # (the three-product forward writes no embedded-input plane: the image-index block is a per-ray bias there;
            #  _attach_explicit makes the plane from the points, mf_nof_embed_rows)
            nstride = A.nof_dump_stride(nof_models[0])
            a.dump_nof_acts = alloc("nof_acts", (steps, N * S, nstride))
            a.dump_nof_stride = nstride
            if x3:
                out["nof_emb"] = None
            else:
                a.dump_nof_emb = alloc("nof_emb", (steps, N * S, 80))
            a.dump_nof_out = alloc("nof_out", (steps, N * S, 3))
    need = int(L.lib().mf_render_workspace_bytes(C.byref(a)))      # bf16 + NoF: the per-ray bias table (ABI v12)
    with torch.cuda.device(dev):
        if need > 0:
            shared = workspace if workspace is not None else [None]
            fresh = shared[0] is None or shared[0].numel() < need
            if fresh:
                shared[0] = torch.empty(need, dtype=torch.uint8, device=dev)
            a.workspace, a.workspace_bytes = shared[0].data_ptr(), need
            keep.append(shared[0])
            if fresh:
                L.check(L.lib().mf_render_prepare(C.byref(a), L.current_stream(dev)), "mf_render_prepare")
        L.check(L.lib().mf_render_pass(C.byref(a), L.current_stream(dev)), "mf_render_pass")
    del keep
    return out


def _loss_partials_hip(c, f, target, N):
    """mf_loss_partials over the arrays the passes wrote: 12 float64 on the device, no host sync."""
    dev = target.device
    lib = L.lib()
    tgt = target.detach().contiguous().float()
    if tgt.shape != (N, 3):
        raise RuntimeError(f"_loss_target must be (N, 3) = ({N}, 3), got {tuple(tgt.shape)}")

    def desc(p):
        d = L.mf_loss_pass()
        d.rgb = L.ptr(p.get("rgb"))
        planes = p.get("disp_local") is not None or p.get("disp_global") is not None
        d.alphas = L.ptr(p.get("alphas")) if planes else None
        d.disp_local, d.disp_global = L.ptr(p.get("disp_local")), L.ptr(p.get("disp_global"))
        d.n_samples = p["alphas"].shape[1] if planes else 0
        return d

    dc = desc(c)
    df = desc(f) if f is not None else None
    out = torch.empty(12, dtype=torch.float64, device=dev)
    scratch = torch.empty(int(lib.mf_loss_partials_scratch_bytes()), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        L.check(lib.mf_loss_partials(C.byref(dc), C.byref(df) if df is not None else None, tgt.data_ptr(), N,
                                     out.data_ptr(), None, scratch.data_ptr(), L.current_stream(dev)), "mf_loss_partials")
    return out


def _compact(alphas, vals_a, vals_b):
    """rendering.py:306-314: (vals[mask] for mask = alphas >= 0.01, all-true if empty)."""
    dev = alphas.device
    N, S = alphas.shape
    lib = L.lib()
    scratch = torch.empty(int(lib.mf_compact_scratch_bytes(N)), dtype=torch.uint8, device=dev)
    count = torch.zeros(1, dtype=torch.int64, device=dev)
    oa = torch.empty(N * S, device=dev, dtype=torch.float32) if vals_a is not None else None
    ob = torch.empty(N * S, device=dev, dtype=torch.float32) if vals_b is not None else None
    with torch.cuda.device(dev):
        L.check(lib.mf_compact_mask(L.ptr(alphas), L.ptr(vals_a), L.ptr(vals_b), N, S, L.ptr(oa), L.ptr(ob),
                                    count.data_ptr(), scratch.data_ptr(), L.current_stream(dev)), "mf_compact_mask")
    n = int(count.item())   # data-dependent length, as in the reference's boolean indexing
    return (oa[:n] if oa is not None else None), (ob[:n] if ob is not None else None)


def _pass_stats(p, N):
    """(sum, count) of the masked consensus distances of ONE pass as device scalars: mf_loss_partials on its planes."""
    dev = p["alphas"].device
    lib = L.lib()
    d = L.mf_loss_pass()
    d.alphas, d.disp_local, d.disp_global = L.ptr(p["alphas"]), L.ptr(p.get("disp_local")), L.ptr(p.get("disp_global"))
    d.n_samples = p["alphas"].shape[1]
    out = torch.empty(12, dtype=torch.float64, device=dev)
    means = torch.empty(6, dtype=torch.float32, device=dev)
    scratch = torch.empty(int(lib.mf_loss_partials_scratch_bytes()), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        L.check(lib.mf_loss_partials(C.byref(d), None, None, N, out.data_ptr(), means.data_ptr(), scratch.data_ptr(),
                                     L.current_stream(dev)), "mf_loss_partials")
    # (sum, count, mean): the mean is a view of the kernel's own fp32 output -- torch.mean(vector) launches nothing
    return {"local": (out[4], out[5], means[2]), "global": (out[8], out[9], means[4]), "out12": out}


def _consensus_vectors(p, N, loc, glob):
    """nof_local_disp / nof_global_disp of a gradient-free pass: lazy.MaskedVector over the kernel's planes (or, with
    LAZY_CONSENSUS off, the compacted tensors right away)."""
    if not LAZY_CONSENSUS:
        return _compact(p["alphas"], p.get("disp_local"), p.get("disp_global"))
    from .lazy import ConsensusPass, MaskedVector

    def compact():
        la, ga = _compact(p["alphas"], p.get("disp_local"), p.get("disp_global"))
        return {"local": la, "global": ga}

    g = ConsensusPass(p["alphas"], {"local": p.get("disp_local"), "global": p.get("disp_global")}, lambda: _pass_stats(p, N), compact,
                      differentiable=False)
    return (MaskedVector(g, "local") if loc else None), (MaskedVector(g, "global") if glob else None)


def sample_pdf(bins, weights, N_importance, det=False, eps=1e-5):
    """Reference signature (rendering.py:5-46): bins (N, nb), weights (N, nb-1) -> (N, N_importance).
    (render_rays itself uses the fused form, ``resample_merge``.)"""
    L.require_gpu(bins, "sample_pdf")
    dev = bins.device
    N, nb = bins.shape
    if weights.shape != (N, nb - 1):
        raise RuntimeError(f"sample_pdf: weights must be (N, {nb - 1}), got {tuple(weights.shape)}")
    M = N_importance
    if det:
        u, u_stride = _linspace01(M, dev), 0
    else:
        u, u_stride = torch.rand(N, M, device=dev), M
    b = bins.detach().contiguous().float()
    w = weights.detach().contiguous().float()
    out = torch.empty((N, M), device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        L.check(L.lib().mf_sample_pdf_eps(L.ptr(b), None, L.ptr(w), nb - 1, N, nb, M, L.ptr(u), u_stride, None,
                                          L.ptr(out), None, None, float(eps), L.current_stream(dev)), "mf_sample_pdf")
    return out


def resample_merge(z_vals, weights, N_importance, det=True, u=None, return_aux=False, cdf=None):
    """sample_pdf(z_mid, weights[:,1:-1]) + cat + sort of render_rays (rendering.py:321-326)
    as one HIP launch. z_vals, weights: (N,S). Returns the sorted (N, S+M) depths."""
    L.require_gpu(z_vals, "resample_merge")
    dev = z_vals.device
    N, S = z_vals.shape
    M = N_importance
    if u is None:
        if det:
            u = _linspace01(M, dev)
            u_stride = 0
        else:
            u = torch.rand(N, M, device=dev)
            u_stride = M
    else:
        u = u.contiguous().float()
        u_stride = M if u.dim() == 2 else 0
    if cdf is not None:
        cdf = cdf.detach().contiguous().float()
    z = z_vals.detach().contiguous().float()
    w = weights.detach().contiguous().float()
    z_out = torch.empty((N, S + M), device=dev, dtype=torch.float32)
    inds = torch.empty((N, M), device=dev, dtype=torch.int32) if return_aux else None
    z_new = torch.empty((N, M), device=dev, dtype=torch.float32) if return_aux else None
    with torch.cuda.device(dev):
        L.check(L.lib().mf_sample_pdf(None, L.ptr(z), w.data_ptr() + 4, S, N, S - 1, M, L.ptr(u), u_stride,
                                      L.ptr(cdf), L.ptr(z_new), L.ptr(inds), L.ptr(z_out),
                                      L.current_stream(dev)), "mf_sample_pdf")
    if return_aux:
        return z_out, inds, z_new
    return z_out


def render_rays(rays,
                background,
                nerf_embeddings,
                nerf_models,
                nof_embeddings=None,
                nof_models=None,
                chain_local=False,
                chain_global=False,
                N_samples=64,
                N_importance=0,
                use_disp=False,
                perturb=0,
                noise_std=1,
                nerf_activate_type='relu',
                test_time=False,
                _capture=None,
                _rng=None,
                _loss_target=None,
                ):
    """Same contract as the reference's render_rays (rendering.py:195-375): rays (N, 9|10),
    background (N,3)|None -> dict with rgb/depth/opacity_{coarse,fine} and, in training with NoF,
    nof_{local,global}_disp_{coarse,fine}. ``_capture`` (dict, test hook) receives the per-pass
    (N,S) planes the kernels produced (z, weights, alphas) without changing the result; ``_rng`` (dict,
    test hook) supplies the random draws instead of torch.rand / randn: perturb_rand (N,S),
    noise_coarse (N,S), noise_fine (N,S+M) (already scaled by noise_std), u (N,M).

    ``_loss_target`` (N,3), the fast path of the mean-only caller (trainer_moco_flow.py:317-328 takes
    ``torch.mean`` of each consensus vector right away): the data-dependent-length ``nof_*_disp_*`` vectors are NOT
    built (no mask compaction, no host sync); instead the result carries ``loss_partials``, 12 float64 on the device
    = (sum, count) of MSE coarse / fine, nof_local coarse / fine, nof_global coarse / fine (dist.loss_partials
    layout; mf_loss_partials), differentiable in training.  ``dist.reduce_loss`` / ``losses.from_partials`` turn
    them into the reference's loss terms."""
    _rng = _rng or {}
    L.require_gpu(rays, "render_rays")
    if nerf_activate_type == 'relu':
        act = L.MF_ACT_RELU
    elif nerf_activate_type == 'softplus':
        act = L.MF_ACT_SOFTPLUS
    else:
        raise ValueError('activation layer type: %s not support' % nerf_activate_type)   # rendering.py:174
    use_nof = nof_models is not None
    if use_nof and chain_global and not chain_local and not test_time:
        # rendering.py:276-280: fw_nof is only bound under chain_local
        raise UnboundLocalError("local variable 'fw_nof' referenced before assignment")
    all_models = list(nerf_models) + (list(nof_models) if use_nof else [])
    grad = A.needs_grad(all_models)       # training call: dump + HIP backward nodes (autograd.py)
    dev = rays.device
    rays = rays.detach().float()
    if rays.dim() != 2 or rays.shape[1] < (10 if (use_nof and chain_global) else 9):
        raise RuntimeError(f"render_rays: rays must be (N, 9|10), got {tuple(rays.shape)}")
    if rays.stride(1) != 1:
        rays = rays.contiguous()
    if background is not None:
        background = background.detach().float().contiguous()
    N = rays.shape[0]
    S = N_samples
    loc = bool(use_nof and chain_local and not test_time)
    glob = bool(use_nof and chain_global and not test_time)
    need_fine = N_importance > 0

    z_steps = _linspace01(S, dev)                                     # rendering.py:245
    z_vals = None
    if perturb > 0 or need_fine or grad:
        # rendering.py:245-251 and, with perturb > 0, the stratified jitter of :253-260: one launch (mf_z_vals), the
        # uniform draws made here with torch in the reference's order
        pr = None
        if perturb > 0:
            pr = _rng["perturb_rand"] if "perturb_rand" in _rng else torch.rand((N, S), device=dev)
            pr = pr.contiguous().float()
        z_vals = torch.empty((N, S), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            L.check(L.lib().mf_z_vals(L.ptr(rays), rays.stride(0), N, L.ptr(z_steps), S, 1 if use_disp else 0,
                                      L.ptr(pr), float(perturb), L.ptr(z_vals), L.current_stream(dev)), "mf_z_vals")

    def draw_noise(shape, key=None):
        if key in _rng:
            return _rng[key].contiguous().float()
        if noise_std != 0:
            return (torch.randn(shape, device=dev) * noise_std).contiguous()
        if STRICT_RNG and shape[0] > 0:
            torch.randn(shape, device=dev)
        return None

    coarse_opacity_only = bool(need_fine and test_time)                # rendering.py:290-294: only opacity_coarse comes back
    # ... from a sigma-only evaluation of the coarse NeRF -- unless the pass records gradients (the reference renders
    # test_time passes under no_grad, but its signature allows it): then the full network runs, its dump is complete, and
    # the backward treats it as the sigma path it is (autograd.NerfSamples, sigma_path_only: rgb-branch parameters get None)
    coarse_sigma_only = coarse_opacity_only and not grad
    # a pass that records gradients runs the reference's fp32 arithmetic (or, NeRF-only passes, the three-product kernels:
    # set_train_forward_precision)
    pass_prec = TRAIN_FORWARD_PRECISION if grad else None
    want_planes = need_fine or loc or glob or grad or _capture is not None
    noise_c = draw_noise((N, S), "noise_coarse")
    if grad and N > 0:
        if coarse_opacity_only and _loss_target is not None:
            raise NotImplementedError("render_rays(test_time=True, N_importance > 0, _loss_target=...) with gradients: the fused loss "
                                      "partials need rgb_coarse, which a test_time pass does not return")
        for m in nerf_models[:2 if need_fine else 1]:
            A.require_nerf_hip(m, N * S, use_nof)
        if use_nof:
            for m in nof_models:
                A.require_nof_hip(m, nof_embeddings)
    ws = [None]                     # bf16 + NoF: the per-ray bias table, filled by the first pass, shared with the second
    c = _render_pass(rays, background, z_vals, None if z_vals is not None else z_steps, use_disp,
                     noise_c, act, nerf_models[0], nerf_embeddings,
                     nof_models if use_nof else None, nof_embeddings, loc, glob, coarse_sigma_only, want_planes,
                     dump=grad and not coarse_sigma_only, precision=pass_prec, workspace=ws)
    if coarse_opacity_only:
        result = {'opacity_coarse': c["opacity"]}
    else:
        result = {'rgb_coarse': c["rgb"], 'depth_coarse': c["depth"], 'opacity_coarse': c["opacity"]}
    fused_loss = _loss_target is not None     # the 12 loss partials replace the compacted consensus vectors
    training = grad and N > 0            # the consensus vectors then come from _attach_explicit, differentiable
    if (loc or glob) and not fused_loss:
        la, ga = (None, None) if training else _consensus_vectors(c, N, loc, glob)
        if loc:
            result['nof_local_disp_coarse'] = la
        if glob:
            result['nof_global_disp_coarse'] = ga

    if _capture is not None:
        _capture.update(z_coarse=z_vals, weights_coarse=c.get("weights"), alphas_coarse=c.get("alphas"))
    if need_fine:
        z_all = resample_merge(z_vals, c["weights"], N_importance, det=(perturb == 0), u=_rng.get("u"))
        noise_f = draw_noise((N, S + N_importance), "noise_fine")
        f = _render_pass(rays, background, z_all, None, use_disp, noise_f, act,
                         nerf_models[1], nerf_embeddings, nof_models if use_nof else None, nof_embeddings,
                         loc, glob, False, loc or glob or grad or _capture is not None,
                         dump=grad, precision=pass_prec, workspace=ws)
        if _capture is not None:
            _capture.update(z_fine=z_all, weights_fine=f.get("weights"), alphas_fine=f.get("alphas"))
        result['rgb_fine'] = f["rgb"]
        result['depth_fine'] = f["depth"]
        result['opacity_fine'] = f["opacity"]
        if (loc or glob) and not fused_loss:
            la, ga = (None, None) if training else _consensus_vectors(f, N, loc, glob)
            if loc:
                result['nof_local_disp_fine'] = la
            if glob:
                result['nof_global_disp_fine'] = ga
    if fused_loss and not (grad and N > 0):
        result['loss_partials'] = _loss_partials_hip(c, f if need_fine else None, _loss_target, N)
    if grad and N > 0:
        result = _attach_explicit(result, rays, background, nerf_embeddings, nerf_models, nof_embeddings,
                                  nof_models if use_nof else None, loc, glob, nerf_activate_type,
                                  (c, z_vals, noise_c), (f, z_all, noise_f) if need_fine else None,
                                  loss_target=_loss_target, coarse_opacity_only=coarse_opacity_only)
    return result


def _attach_explicit(result, rays, background, nerf_embs, nerf_models, nof_embs, nof_models, loc, glob,
                     activation, coarse, fine, loss_target=None, coarse_opacity_only=False):
    """Training graph on top of the fused forward (fp32): values are the HIP
    kernels' outputs; gradients flow through
      * autograd.CompositeSamples -- mf_composite_backward on the dumped per-sample (rgb, sigma) planes,
      * autograd.NerfSamples -- mf_nerf_backward + mf_weight_grads over the kernel's activation dump
        (no recompute of the 12-layer MLP, no library GEMM),
      * autograd.NofPoints -- one HIP forward-with-dump / backward node per NoF evaluation of the chains.
    Each returned tensor is  hip_value + (torch_value - torch_value.detach())."""
    rays_o, rays_d, ind = rays[:, 0:3], rays[:, 3:6], rays[:, 8:9]
    out, recons, kernel_vals = {}, {}, set()

    def one(tag, nerf, pack, opacity_only=False):
        p, z, noise = pack
        N, S = z.shape
        xyz = rays_o.unsqueeze(1) + rays_d.unsqueeze(1) * z.unsqueeze(2)
        group = None
        if (loc or glob) and loss_target is None:
            from .lazy import ConsensusPass, MaskedVector
            planes, recon_of = {}, {}
            # torch.mean(vector) -- all the trainer asks of these entries -- is ONE autograd node on the distances the fused
            # pass wrote (autograd.ConsensusMean); the per-sample planes are only built (torch ops) when something else is asked
            # (the callbacks take the pass as an argument: closing over `group` would tie it into a reference cycle with them,
            #  and the pass -- every dump plane of it -- would live until the cyclic collector runs, lazy.ConsensusPass)
            group = ConsensusPass(p["alphas"], planes, lambda: _pass_stats(p, N),
                                  lambda g: {k: torch.masked_select(g.plane(k), _mask_of(p["alphas"])) for k in planes}, True,
                                  mean_fn=(lambda g, k: A.ConsensusMean.apply(g, k, p, rays, z, recon_of[k])) if FUSED_CONSENSUS_MEAN else None,
                                  pass_self=True)

            def vector(key, recon):
                # torch.mean(|xyz - recon|[mask], dim=1) of rendering.py:310-314 as mean-then-select (same numbers; the backward
                # is a masked scatter instead of the sort + accumulate of boolean-index backward), lazily (lazy.MaskedVector)
                recon_of[key] = recon
                planes[key] = lambda: torch.abs(xyz - recon).mean(-1)
                v = MaskedVector(group, key)
                return v if LAZY_CONSENSUS else v.materialize()
        xin = p["xyz_in"]
        if nof_models is not None:
            bw = nof_models[0]
            if "nof_acts" in p:       # the fused pass dumped every evaluation of the chain: nodes without a forward launch
                step = [0]
                plane = p["nof_plane"]
                if p.get("nof_emb") is None:
                    p["nof_emb"] = _nof_embedded_inputs(p, plane, xyz.detach().reshape(-1, 3), rays, S, nof_embs)
                n_bw = sum(1 for k in range(len(plane)) if k in (0, 3))
                # per network: its planes are adjacent; with whole 128-row blocks per plane the nodes leave their
                # pre-activation gradients in one buffer and ONE mf_weight_grads launch per network follows them
                # (the sinks are keyed by module: not when one NoF object plays both roles; and a frozen network has no
                #  weight gradients to batch -- its gate's backward would never run and flush the buffer)
                batched = (N * S) % 128 == 0 and (len(nof_models) < 2 or nof_models[0] is not nof_models[1])
                trains = lambda m: any(q.requires_grad for q in m.parameters())
                sinks = {}
                slots = not getattr(p["nof_emb"], "_mf_natural", False)
                if batched and trains(bw):
                    sinks[id(bw)] = (A.NofGradSink(bw, p["nof_acts"][:n_bw], p["nof_emb"][:n_bw], slots), 0)
                if batched and len(plane) > n_bw and trains(nof_models[1]):
                    sinks[id(nof_models[1])] = (A.NofGradSink(nof_models[1], p["nof_acts"][n_bw:], p["nof_emb"][n_bw:], slots), n_bw)
                gated = {key: A.NofParamGate.apply(sk, *sk.m.parameters()) for key, (sk, _) in sinks.items()}

                def nof_points(pts, ray_ind, embs_, m):
                    k = plane[step[0]]
                    step[0] += 1
                    sk, first = sinks.get(id(m), (None, 0))
                    emb_k = p["nof_emb"][k]
                    emb_k._mf_natural = not slots
                    return A.nof_points_dumped(pts, nof_embs, m, p["nof_acts"][k], emb_k, p["nof_out"][k],
                                               sink=sk, sink_plane=k - first, params=gated.get(id(m)))
            else:
                nof_points = A.nof_points
            canon = nof_points(xyz, ind, nof_embs, bw)
            if loc:
                fw = nof_models[1]
                recon = nof_points(canon, ind, nof_embs, fw)
                if loss_target is None:
                    out[f"nof_local_disp_{tag}"] = vector("local", recon)
                else:
                    recons[f"local_{tag}"] = recon           # the loss node differentiates |x - recon| itself
            if glob:
                cind = rays[:, 9:10]
                a_ = nof_points(canon, cind, nof_embs, fw)
                b_ = nof_points(a_, cind, nof_embs, bw)
                chained = nof_points(b_, ind, nof_embs, fw)
                if loss_target is None:
                    out[f"nof_global_disp_{tag}"] = vector("global", chained)
                else:
                    recons[f"global_{tag}"] = chained
            xin = canon.reshape(-1, 3)
        with torch.no_grad():
            # (mf_embedding_forward_rows: the 64- / 32-column operands of the first-layer weight gradients, the per-ray
            #  block repeated for the ray's samples, zero padded -- no repeat_interleave / pad copies)
            emb_in = nerf_embs[0].rows(p["xyz_in"], 1, max(64, nerf.in_channels_xyz))
            extra_in = None
            if nerf.extra_feat_type == "ind":
                extra_in = nerf_embs[1].rows(ind, S, max(32, nerf.extra_feat_dim))
            elif nerf.extra_feat_type == "dir":
                extra_in = nerf_embs[2].rows(rays_d, S, max(32, nerf.extra_feat_dim))
        rgbsig = A.NerfSamples.apply(nerf, p["acts"], p["rgbsig"], emb_in, extra_in, nerf_embs[0], xin, opacity_only,
                                     *nerf.parameters())
        if S > 2048:
            raise NotImplementedError(f"render_rays with gradients: mf_composite_backward is built for <= 2048 samples per ray, got {S}")
        out[f"rgb_{tag}"], out[f"depth_{tag}"], out[f"opacity_{tag}"] = A.CompositeSamples.apply(
            rgbsig, rays, z, noise, activation, background, p["rgb"], p["depth"], p["opacity"])
        kernel_vals.update((f"rgb_{tag}", f"depth_{tag}", f"opacity_{tag}"))

    one("coarse", nerf_models[0], coarse, opacity_only=coarse_opacity_only)
    if fine is not None:
        one("fine", nerf_models[1], fine)
    final = {}
    for k, v in result.items():
        t = out[k]
        # CompositeSamples hands back the kernel's own values (detached clones), the consensus vectors are built here
        final[k] = t if (k in kernel_vals or v is None) else v.detach() + (t - t.detach())
    if loss_target is not None:
        # the 12 (sum, count) partials (layout of dist.loss_partials / mf_loss_partials) as one autograd node: forward =
        # the kernel on the planes the passes wrote, backward = one launch writing the seeds (autograd.LossPartials)
        tgt = loss_target.detach().float().contiguous()
        passes = [dict(planes=coarse[0], z=coarse[1])] + ([dict(planes=fine[0], z=fine[1])] if fine is not None else [])
        tensors = []
        for tag in ("coarse", "fine")[:len(passes)]:
            tensors += [out[f"rgb_{tag}"], recons.get(f"local_{tag}"), recons.get(f"global_{tag}")]
        N = rays.shape[0]
        final["loss_partials"] = A.LossPartials.apply(
            lambda: _loss_partials_hip(coarse[0], fine[0] if fine is not None else None, tgt, N), rays, tgt, passes, *tensors)
    return final


def _nof_embedded_inputs(p, plane, x_obs, rays, S, nof_embs):
    """(steps, N S, 80) embedded inputs of the chain's NoF evaluations in natural column order (mf_nof_embed_rows), from the
    points each step saw: step 0 the observation-space samples, then the dumped outputs of the steps in front
    (rendering.py:270-282: bw(x, i); fw(canon, i); fw(canon, j); bw(., j); fw(., i))."""
    steps, P = len(plane), x_obs.shape[0]
    out_of = lambda k: p["nof_out"][plane[k]]
    src = {0: x_obs, 1: out_of(0), 2: out_of(0)}
    if steps > 3:
        src[3], src[4] = out_of(2), out_of(3)
    if steps == 2:
        src = {0: x_obs, 1: out_of(0)}
    emb = torch.empty((steps, P, 80), device=x_obs.device, dtype=torch.float32)
    ex = nof_embs[0].descriptor()
    with torch.no_grad():                                   # the index block is constant along a ray: embedded once per ray
        ind_i = nof_embs[1](rays[:, 8:9].contiguous())
        ind_j = nof_embs[1](rays[:, 9:10].contiguous()) if steps > 2 else None
    with torch.cuda.device(x_obs.device):
        for k in range(steps):
            tab = ind_j if k in (2, 3) else ind_i
            pts = src[k].contiguous()
            L.check(L.lib().mf_nof_embed_rows(ex, L.ptr(pts), L.ptr(tab), tab.shape[1], S, P, emb[plane[k]].data_ptr(),
                                              L.current_stream(x_obs.device)), "mf_nof_embed_rows")
    emb._mf_natural = True
    return emb


def _mask_of(alphas):
    mask = alphas.ge(0.01)                      # rendering.py:306-308
    if not torch.any(mask):
        mask = torch.ones_like(mask).bool()
    return mask
