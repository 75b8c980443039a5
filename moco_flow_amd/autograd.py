"""Backward pass of the drop-in modules.

Forward values always come from the fused HIP kernels.  When gradients are requested the fp32 forward additionally
DUMPS what the backward needs (per-sample post-activation layer outputs, rgb / sigma, NeRF input points, every NoF
evaluation of the chains; mf_render_args.dump_*) and the backward is HIP too, as a handful of autograd nodes whose
backward methods are single launches:
    ``LossPartials``      mf_loss_partials / mf_loss_partials_backward   the 12 (sum, count) loss partials of the step
    ``CompositeSamples``  mf_composite_backward   dL/d(rgb, depth, opacity) -> dL/d(rgb, sigma) per sample
    ``NerfSamples``       mf_nerf_backward_x + mf_weight_grads   the 12-layer NeRF: input-gradient chain on the transposed
                          weights over the dump, then every dW / db in one persistent launch
    ``NofPointsDumped``   mf_nof_backward (+ one mf_weight_grads per network, NofGradSink)   one NoF evaluation of a chain
    ``NofPoints``         the same node with its own dumping forward (mf_nof_points_dump)
``NeRF(x[, sigma_only])`` / ``NoF(x, xyz)`` / ``Embedding(x)`` called directly (the joint stage's point losses):
``NerfModule`` / ``NofModule`` / ``EmbeddingModule`` -- a dumping forward launch, then the same backward launches.
No forward recompute, no autograd graph over any MLP, no eager restatement of the reference anywhere in the package: a
shape the HIP backward is not built for raises NotImplementedError under grad (require_nerf_hip / require_nof_hip; the
reference's three YAML configurations are inside the envelope).  The eager op sequence the tools time against lives in
tools/eager_ref.py.

Gradients reach every parameter that requires grad (frozen sub-modules are honoured, trainer_moco_flow.py:391-404) and do
not flow through the resampled depths (rendering.py:323).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from . import _lib as L


# ------------------------------------------------------------------ autograd glue
def needs_grad(modules) -> bool:
    return torch.is_grad_enabled() and any(p.requires_grad for m in modules if m is not None for p in m.parameters())


# ------------------------------------------------------------------ explicit NeRF backward on the kernel's dump
# Arithmetic of the NeRF's input-gradient chain (the D + 1 W-wide contractions on the transposed weights): "f32" =
# mf_nerf_backward_x; "bf16x3" = mf_nerf_backward3 (three bf16 products of (hi, lo) pairs, masks from the dump as in fp32,
# incl. the gradient of the embedded input the joint stage's NoFs need).  End-to-end
# gradients against the oracle's autograd: 8e-6 max-rel (fp32 chain: 2e-6; bar 1e-4); stage-1 step 47.4 -> 41.6 ms.  The
# default; set_dx_precision("f32") restores the exact-fp32 chain.
DX_PRECISION = "bf16x3"


def set_dx_precision(p: str):
    global DX_PRECISION
    if p not in ("f32", "bf16x3"):
        raise ValueError(f"dX precision must be 'f32' or 'bf16x3', got {p!r}")
    DX_PRECISION = p


def nof_dump_stride(m):
    """Floats per row of a NoF evaluation's dump: [h_1 .. h_D | T padded to 16 | the layers' ReLU bit rows, 4 words each].  The
    bit rows are what mf_nof_backward3 reads instead of the activations (the forward writes them when the row has room)."""
    return m.D * m.W + 16 + 4 * m.D


def nof_backward_hip(m, emb_desc, P, pts, acts, stride, g_out, gpre, g_pts):
    """One NoF evaluation's backward launch: mf_nof_backward3 (three bf16 products, set_dx_precision("bf16x3"), the default)
    or mf_nof_backward (exact-fp32 MFMA).  Same arguments, same outputs (gpre rows + the point gradient)."""
    dev = pts.device
    # (the three-product chain is built for the full 33 + 33 input block, mf_nofgrad_bf16.hip nof_bwd3_shape; anything else, and
    #  misaligned dumps, take the exact-fp32 kernel)
    x3 = (DX_PRECISION == "bf16x3" and stride % 4 == 0 and acts.data_ptr() % 16 == 0 and gpre.data_ptr() % 16 == 0
          and m.in_channels_xyz + m.extra_feat_dim >= 64)
    desc, buf = m.packed_bwd3() if x3 else m.packed_bwd()
    fn, what = (L.lib().mf_nof_backward3, "mf_nof_backward3") if x3 else (L.lib().mf_nof_backward, "mf_nof_backward")
    with torch.cuda.device(dev):
        L.check(fn(C.byref(desc), buf.data_ptr(), C.byref(emb_desc), P, pts.data_ptr(), acts.data_ptr(), stride,
                   g_out.data_ptr(), gpre.data_ptr(), L.ptr(g_pts), L.current_stream(dev)), what)


def nerf_backward_hip(m, g_out, acts, rgbsig, want_emb=False):
    """mf_nerf_backward_x: (gpre (P,stride) in the dump's layout, ghead (P,4), g_emb (P,64) | None) from
    dL/d[rgb, sigma]; g_emb = the gradient of the embedded input, produced by the same launch."""
    P, stride = acts.shape
    dev = acts.device
    g_out = g_out.contiguous().float()
    gpre = torch.empty(((P + 127) // 128 * 128, stride), device=dev, dtype=torch.float32)
    ghead = torch.empty((P, 4), device=dev, dtype=torch.float32)
    if DX_PRECISION == "bf16x3" and m.W == 256 and m.D >= 2 and stride % 4 == 0 and (not want_emb or len(m.skips) <= 1):
        desc, buf = m.packed_bwd3()
        g_emb = torch.empty((P, 64), device=dev, dtype=torch.float32) if want_emb else None
        mask = getattr(acts, "_mf_mask", None)          # the forward's ReLU bit-mask rows, when it wrote them (rendering.py)
        if mask is not None and (mask.shape[0] != P or mask.shape[1] < (m.D + 2) * 8):
            mask = None
        with torch.cuda.device(dev):
            L.check(L.lib().mf_nerf_backward3(C.byref(desc), buf.data_ptr(), P, g_out.data_ptr(), acts.data_ptr(), stride,
                                              rgbsig.data_ptr(), gpre.data_ptr(), ghead.data_ptr(), L.ptr(g_emb),
                                              L.ptr(mask), mask.shape[1] if mask is not None else 0,
                                              L.current_stream(dev)), "mf_nerf_backward3")
        return gpre[:P], ghead, g_emb
    desc, buf = m.packed_bwd()
    g_emb = torch.empty((P, 64), device=dev, dtype=torch.float32) if want_emb else None
    with torch.cuda.device(dev):
        L.check(L.lib().mf_nerf_backward_x(C.byref(desc), buf.data_ptr(), P, g_out.data_ptr(), acts.data_ptr(), stride,
                                           rgbsig.data_ptr(), gpre.data_ptr(), ghead.data_ptr(), L.ptr(g_emb),
                                           L.current_stream(dev)), "mf_nerf_backward")
    return gpre[:P], ghead, g_emb


def embed_backward_hip(emb, emb_vals, g_emb):
    """mf_embedding_backward: d/dx of embedding.py:42-46 through the embedded values (their sin / cos columns already
    carry the per-frequency weights): (P, C) from g_emb (P, >= width), emb_vals (P, >= width)."""
    P, dev = g_emb.shape[0], g_emb.device
    desc = emb.descriptor()
    g_x = torch.empty((P, emb.in_channels), device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        L.check(L.lib().mf_embedding_backward(C.byref(desc), g_emb.data_ptr(), g_emb.stride(0), emb_vals.data_ptr(),
                                              emb_vals.stride(0), P, g_x.data_ptr(), L.current_stream(dev)),
                "mf_embedding_backward")
    return g_x


class EmbeddingModule(torch.autograd.Function):
    """``Embedding(x)`` with x requiring grad (embedding.py:30-47; trainer_moco_flow.py:147-149): forward
    mf_embedding_forward, backward mf_embedding_backward on the saved embedded values."""

    @staticmethod
    def forward(ctx, emb, x):
        xc = x.detach().contiguous().float()
        out = torch.empty((xc.shape[0], emb.out_channels), device=xc.device, dtype=torch.float32)
        d = emb.descriptor()
        with torch.cuda.device(xc.device):
            L.check(L.lib().mf_embedding_forward(d, L.ptr(xc), xc.shape[0], L.ptr(out), L.current_stream(xc.device)),
                    "mf_embedding_forward")
        ctx.emb = emb
        ctx.save_for_backward(out)
        return out.clone()

    @staticmethod
    def backward(ctx, g):
        out, = ctx.saved_tensors
        if out.shape[0] == 0:
            return None, g.new_zeros((0, ctx.emb.in_channels))
        return None, embed_backward_hip(ctx.emb, out, g.contiguous().float())


_WG_BLOCK = {(256, 256): (256, 256), (256, 64): (256, 64), (128, 256): (128, 256), (128, 32): (128, 32), (4, 640): (16, 640),
             (128, 128): (128, 128), (128, 80): (128, 80), (12, 128): (16, 128)}


# Arithmetic of the weight-gradient contractions dW = G^T X (mf_weight_grads_p): "f32" = exact-fp32 MFMA; "bf16x3" = every
# block but the heads' 4 x 640 (round 5: the NeRF's 256x256, 128x256, 256x64, 128x32 and the NoF's 128x128, 128x80, 12x128) as
# three bf16 products of (hi, lo) operand pairs with fp32 accumulation (include/mocoflow_hip.h): 16 mantissa bits per operand,
# a sum over ~1e6 samples -- 6e-6 .. 1.6e-5 l2-rel against a float64 GEMM where the fp32 MFMA measures 4e-6
# (tools/bench_wgrad.py), every gradient test holds its bar in both (tests/conftest.py `wgrad`) -- at 0.56x (NeRF) / 0.61x (NoF)
# the time of the fp32 launch: it runs against the HBM reads of its operands.  The default; set_wgrad_precision("f32")
# restores the exact-fp32 contraction.
WGRAD_PRECISION = "bf16x3"


def set_wgrad_precision(p: str):
    global WGRAD_PRECISION
    if p not in ("f32", "bf16x3"):
        raise ValueError(f"wgrad precision must be 'f32' or 'bf16x3', got {p!r}")
    WGRAD_PRECISION = p


_WGRAD_ARRAY_TYPES = {}


def _wgrad_items(n):
    """The ctypes array type of n items, made once per n (every `mf_wgrad_item * n` is a NEW type object, and type objects are
    cyclic garbage: a few per training step for the collector)."""
    t = _WGRAD_ARRAY_TYPES.get(n)
    if t is None:
        t = _WGRAD_ARRAY_TYPES[n] = L.mf_wgrad_item * n
    return t


def weight_grads(jobs, P, dev):
    """mf_weight_grads_p: jobs = [(G, X, n_out, n_in, want_bias)] with G / X fp32 device matrices (column
    slices allowed) -> [(dW (rows, n_in), db (rows,) | None)] in ONE persistent HIP launch (two when WGRAD_PRECISION is
    "bf16x3": the blocks with a three-product variant and the rest)."""
    n = len(jobs)
    if n == 0:
        return []
    if n > L.MF_WG_MAX_ITEMS:
        raise RuntimeError(f"weight_grads: {n} items (max {L.MF_WG_MAX_ITEMS})")
    items = _wgrad_items(n)()
    outs = []
    for it, (G, X, n_out, n_in, bias) in zip(items, jobs):
        rows, cols = _WG_BLOCK[(n_out, n_in)]
        dW = torch.empty((rows, cols), device=dev, dtype=torch.float32)
        db = torch.empty((rows,), device=dev, dtype=torch.float32) if bias else None
        if G.stride(1) != 1 or X.stride(1) != 1:
            raise RuntimeError("weight_grads: operands must be row-major")
        it.G, it.g_stride, it.n_out = G.data_ptr(), G.stride(0), n_out
        it.X, it.x_stride, it.n_in = X.data_ptr(), X.stride(0), n_in
        it.dW, it.db = dW.data_ptr(), (db.data_ptr() if bias else None)
        outs.append((dW, db))
    lib = L.lib()
    prec = L.PRECISIONS[WGRAD_PRECISION]
    nbytes = lib.mf_weight_grads_scratch_bytes_p(prec, items, n, P)
    if nbytes < 0:
        L.check(-3, "mf_weight_grads")
    scratch = torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        L.check(lib.mf_weight_grads_p(prec, items, n, P, scratch.data_ptr(), L.current_stream(dev)), "mf_weight_grads")
    return outs


# ------------------------------------------------------------------ NoF evaluation on points, HIP forward + backward
def nof_hip_supported(m, nof_embs) -> bool:
    """nof_embs = None: module-level call on pre-embedded inputs (no embedding constraints)."""
    skips = [s for s in m.skips if 0 < s < m.D]
    ok = m.W == 128 and m.in_channels_xyz == 33 and m.extra_feat_dim == 33 and len(skips) <= 1 and 2 <= m.D <= 8
    return ok and (nof_embs is None or (nof_embs[0].N_freqs <= 5 and nof_embs[1].N_freqs == 16))


_SLOT_COLS = {}


def _nof_slot_columns(dev):
    """For each of the 66 embedded-input columns, its column in the slot-ordered dump of the fused pass
    (mf_render_args.dump_nof_emb, map from mf_nof_emb_slot_features)."""
    key = str(dev)
    if key not in _SLOT_COLS:
        feats = (C.c_int32 * 80)()
        L.check(L.lib().mf_nof_emb_slot_features(feats), "mf_nof_emb_slot_features")
        where = [0] * 66
        for c, f in enumerate(feats):
            if f >= 0:
                where[f] = c
        _SLOT_COLS[key] = torch.tensor(where, dtype=torch.int64, device=dev)
    return _SLOT_COLS[key]


def _nof_param_grads(m, gpre, acts, emb80, req, slot_order=False):
    """dW / db of every NoF layer from the gradient buffer of mf_nof_backward and the forward dump:
    ONE mf_weight_grads launch (emb80 = the embedded input, 66 columns padded to 80; ``slot_order``: its columns are in
    the fused pass' register-slot order and the 128 x 80 result is gathered back)."""
    P, dev = acts.shape[0], acts.device
    D, W = m.D, m.W
    names = [n for n, _ in m.named_parameters()]
    grads = {n: None for n in names}
    gslot = lambda l: gpre[:, l * W:(l + 1) * W]
    h = lambda l: acts[:, l * W:(l + 1) * W]
    cin = m.in_channels_xyz + m.extra_feat_dim
    wants = lambda prefix: req[prefix + ".weight"] or req[prefix + ".bias"]
    jobs, sinks = [], []
    for l in range(D):
        name = f"nof_encoding_{l+1}.0"
        if not wants(name):
            continue
        blocks = []
        if l == 0 or l in m.skips:
            jobs.append((gslot(l), emb80, 128, 80, l == 0))
            blocks.append((len(jobs) - 1, _nof_slot_columns(dev)[:cin] if slot_order else slice(0, cin)))
        if l > 0:
            jobs.append((gslot(l), h(l - 1), 128, 128, True))
            blocks.append((len(jobs) - 1, slice(0, W)))
        sinks.append((name, blocks, blocks[-1][0]))
    head_job = None
    if wants("nof_encoding_final"):
        jobs.append((gpre[:, D * W:D * W + 12], h(D - 1), 12, 128, True))
        head_job = len(jobs) - 1
    res = weight_grads(jobs, P, dev)
    for prefix, blocks, bias_from in sinks:
        if req[prefix + ".weight"]:
            parts = [res[j][0][:, cs] for j, cs in blocks]
            grads[prefix + ".weight"] = parts[0] if len(parts) == 1 else torch.cat(parts, 1)
        if req[prefix + ".bias"]:
            grads[prefix + ".bias"] = res[bias_from][1]
    if head_job is not None:
        nh = m.nof_encoding_final.weight.shape[0]
        if req["nof_encoding_final.weight"]:
            grads["nof_encoding_final.weight"] = res[head_job][0][:nh]
        if req["nof_encoding_final.bias"]:
            grads["nof_encoding_final.bias"] = res[head_job][1][:nh]
    return grads


class NofModule(torch.autograd.Function):
    """NoF.forward(inputs, xyz) (models/nof.py:55-85) as called directly by the trainers
    (trainer_nof.py:85-112 -- the whole of stage 2 --, trainer_moco_flow.py:159-187) when only the
    parameters need gradients: forward mf_nof_forward_dump, backward mf_nof_backward + mf_weight_grads."""

    @staticmethod
    def forward(ctx, m, inputs, xyz, *params):
        B, dev = inputs.shape[0], inputs.device
        desc, buf = m.packed()
        stride = nof_dump_stride(m)
        x = inputs.detach().float()
        if x.stride(1) != 1:
            x = x.contiguous()
        pts = xyz.detach().float().contiguous()
        out = torch.empty((B, 3), device=dev, dtype=torch.float32)
        acts = torch.empty((B, stride), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            L.check(L.lib().mf_nof_forward_dump(C.byref(desc), buf.data_ptr(), x.data_ptr(), x.stride(0), pts.data_ptr(), B,
                                                out.data_ptr(), acts.data_ptr(), stride, L.current_stream(dev)),
                    "mf_nof_forward_dump")
        ctx.m, ctx.stride = m, stride
        ctx.save_for_backward(x, pts, acts)
        return out

    @staticmethod
    def backward(ctx, g_out):
        m, stride = ctx.m, ctx.stride
        x, pts, acts = ctx.saved_tensors
        B, dev = x.shape[0], x.device
        names = [n for n, _ in m.named_parameters()]
        req = {n: p.requires_grad for n, p in m.named_parameters()}
        with torch.no_grad():
            noemb = L.mf_embedding()
            noemb.in_channels, noemb.n_freqs = 3, 0               # no point gradient is requested
            g_out = g_out.contiguous().float()
            gpre = torch.empty(((B + 127) // 128 * 128, stride), device=dev, dtype=torch.float32)
            nof_backward_hip(m, noemb, B, pts, acts, stride, g_out, gpre, None)
            grads = _nof_param_grads(m, gpre[:B], acts, F.pad(x, (0, 80 - x.shape[1])), req)
        return (None, None, None) + tuple(grads[n] for n in names)


class NofPoints(torch.autograd.Function):
    """out = NoF([embed(pts) | embed(ind)], pts) for free points (rendering.py:49-83 + nof.py:69-82) with
    the image index given per RAY (sample s belongs to ray s // S).  Forward: mf_nof_points_dump (the
    fused MFMA core, storing the layer outputs); backward: mf_nof_backward (transform, head, W^T chain,
    embedding chain rule -> d pts) + mf_weight_grads (dW / db of every layer).  The consensus chains of
    render_rays are compositions of this node, so torch only sees 1-5 of them per pass."""

    @staticmethod
    def forward(ctx, m, nof_embs, ray_ind, S, pts, *params):
        P, dev = pts.shape[0], pts.device
        pts = pts.detach().contiguous().float()
        desc, buf = m.packed()
        stride = nof_dump_stride(m)
        out = torch.empty((P, 3), device=dev, dtype=torch.float32)
        acts = torch.empty((P, stride), device=dev, dtype=torch.float32)
        emb = torch.empty((P, 80), device=dev, dtype=torch.float32)
        ex, ei = nof_embs[0].descriptor(), nof_embs[1].descriptor()
        with torch.cuda.device(dev):
            L.check(L.lib().mf_nof_points_dump(C.byref(desc), buf.data_ptr(), C.byref(ex), C.byref(ei), pts.data_ptr(),
                                               ray_ind.data_ptr(), ray_ind.stride(0), S, P, out.data_ptr(),
                                               acts.data_ptr(), stride, emb.data_ptr(), L.current_stream(dev)),
                    "mf_nof_points_dump")
        ctx.m, ctx.ex, ctx.stride = m, ex, stride
        ctx.save_for_backward(pts, acts, emb)
        return out

    @staticmethod
    def backward(ctx, g_out):
        m, stride = ctx.m, ctx.stride
        pts, acts, emb = ctx.saved_tensors
        P, dev = pts.shape[0], pts.device
        D, W = m.D, m.W
        names = [n for n, _ in m.named_parameters()]
        req = {n: p.requires_grad for n, p in m.named_parameters()}
        grads = {n: None for n in names}
        need_pts = ctx.needs_input_grad[4]
        with torch.no_grad():
            g_out = g_out.contiguous().float()
            gpre = torch.empty(((P + 127) // 128 * 128, stride), device=dev, dtype=torch.float32)
            g_pts = torch.empty((P, 3), device=dev, dtype=torch.float32) if need_pts else None
            nof_backward_hip(m, ctx.ex, P, pts, acts, stride, g_out, gpre, g_pts)
            grads = _nof_param_grads(m, gpre[:P], acts, emb, req)
        return (None, None, None, None, g_pts) + tuple(grads[n] for n in names)


class NofGradSink:
    """Weight gradients of ONE NoF over all of its evaluations in a render pass.  The chain's nodes (NofPointsDumped)
    leave their pre-activation gradients in planes of one buffer laid out like the forward dump (the fused pass keeps a
    network's evaluations in adjacent planes); NofParamGate's backward -- which autograd runs after every consumer of
    the gated parameters -- turns all of them into dW / db with one mf_weight_grads launch (instead of one launch and
    one gradient accumulation per evaluation)."""

    def __init__(self, m, acts, emb, slot_order=True):
        self.m, self.acts, self.emb = m, acts, emb           # (n, P, stride), (n, P, 80): this network's planes
        self.slot_order = slot_order                         # columns of `emb`: the fp32 pass' register slots | natural (mf_nof_embed_rows)
        self.gpre, self.filled = None, set()

    def plane(self, k):
        if self.gpre is None:
            self.gpre = torch.empty_like(self.acts)
        self.filled.add(k)
        return self.gpre[k]

    def flush(self):
        m = self.m
        names = [n for n, _ in m.named_parameters()]
        req = {n: p.requires_grad for n, p in m.named_parameters()}
        total = {n: None for n in names}
        ks = sorted(self.filled)
        runs, i = [], 0
        while i < len(ks):                                    # maximal runs of adjacent planes (normally one)
            j = i
            while j + 1 < len(ks) and ks[j + 1] == ks[j] + 1:
                j += 1
            runs.append((ks[i], ks[j] + 1))
            i = j + 1
        for lo, hi in runs:
            flat = lambda t: t[lo:hi].reshape(-1, t.shape[-1])
            g = _nof_param_grads(m, flat(self.gpre), flat(self.acts), flat(self.emb), req, slot_order=self.slot_order)
            for n in names:
                if g[n] is not None:
                    total[n] = g[n] if total[n] is None else total[n] + g[n]
        self.gpre, self.filled = None, set()
        self.acts = self.emb = None                       # (this network's dump planes of the pass: dead now, see NerfSamples.backward)
        return [total[n] for n in names]


class NofParamGate(torch.autograd.Function):
    """Identity on a network's parameters whose backward is NofGradSink.flush (see there)."""

    @staticmethod
    def forward(ctx, sink, *params):
        ctx.sink = sink
        ctx.set_materialize_grads(False)
        return tuple(p.view_as(p) for p in params)

    @staticmethod
    def backward(ctx, *_unused):
        with torch.no_grad():
            return (None,) + tuple(ctx.sink.flush())


class NofPointsDumped(torch.autograd.Function):
    """The same node when the fused render pass already evaluated and dumped this step of the chain
    (mf_render_args.dump_nof_*): forward = the dumped output points, no launch; backward as NofPoints, the weight
    gradients either here or -- with a ``sink`` -- deferred to the network's NofParamGate."""

    @staticmethod
    def forward(ctx, m, nof_embs, acts, emb, out, sink, sink_plane, pts, *params):
        ctx.m, ctx.ex, ctx.stride = m, nof_embs[0].descriptor(), acts.shape[1]
        ctx.sink, ctx.sink_plane = sink, sink_plane
        ctx.slot_order = not getattr(emb, "_mf_natural", False)
        ctx.save_for_backward(pts.detach().contiguous().float(), acts, emb)
        ctx.set_materialize_grads(False)
        return out.detach()                                   # the dump's own plane (nobody writes it after the pass)

    @staticmethod
    def backward(ctx, g_out):
        m, stride = ctx.m, ctx.stride
        pts, acts, emb = ctx.saved_tensors
        P, dev = pts.shape[0], pts.device
        names = [n for n, _ in m.named_parameters()]
        req = {n: p.requires_grad for n, p in m.named_parameters()}
        need_pts = ctx.needs_input_grad[7]
        if g_out is None:
            return (None,) * (8 + len(names))
        with torch.no_grad():
            g_out = g_out.contiguous().float()
            if ctx.sink is not None:
                gpre = ctx.sink.plane(ctx.sink_plane)                 # (P, stride), P % 128 == 0
            else:
                gpre = torch.empty(((P + 127) // 128 * 128, stride), device=dev, dtype=torch.float32)
            g_pts = torch.empty((P, 3), device=dev, dtype=torch.float32) if need_pts else None
            nof_backward_hip(m, ctx.ex, P, pts, acts, stride, g_out, gpre, g_pts)
            if ctx.sink is not None:
                return (None,) * 7 + (g_pts,) + (None,) * len(names)
            grads = _nof_param_grads(m, gpre[:P], acts, emb, req, slot_order=ctx.slot_order)
        return (None,) * 7 + (g_pts,) + tuple(grads[n] for n in names)


def nof_points_dumped(xyz, nof_embs, m, acts, emb, out, sink=None, sink_plane=0, params=None):
    """(N,S,3) points -> the dumped NoF output (N,S,3), differentiable w.r.t. the points and the NoF parameters
    (``params``: the network's NofParamGate outputs when the weight gradients go through ``sink``)."""
    N, S = xyz.shape[:2]
    params = tuple(m.parameters()) if params is None else params
    return NofPointsDumped.apply(m, nof_embs, acts, emb, out, sink, sink_plane, xyz.reshape(-1, 3), *params).view(N, S, 3)


def nof_points(xyz, ray_ind, nof_embs, m):
    """One NoF evaluation of a chain on (N,S,3) points with per-ray indices as a HIP forward / backward node."""
    N, S = xyz.shape[0], xyz.shape[1]
    require_nof_hip(m, nof_embs)
    return NofPoints.apply(m, nof_embs, ray_ind, S, xyz.reshape(-1, 3), *m.parameters()).view(N, S, 3)


def require_nof_hip(m, nof_embs):
    if not nof_hip_supported(m, nof_embs):
        raise NotImplementedError(f"render_rays with gradients: the NoF backward is built for W = 128, in_channels_xyz = 33, "
                                  f"extra_feat_dim = 33, at most one skip layer, xyz embedding <= 5 and index embedding = 16 "
                                  f"frequencies (got W={m.W}, D={m.D}, skips={m.skips}); there is no eager fallback")


def nerf_fused_eligible(m, P):
    """Shapes the fused dX chain + mf_weight_grads launches are built for."""
    D, W = m.D, m.W
    return (P > 0 and W == 256 and m.in_channels_xyz <= 64 and m.extra_feat_dim <= 32
            and D + len([s for s in m.skips if 0 < s < D]) + 4 <= L.MF_WG_MAX_ITEMS)


def nerf_fused_grads(m, g_out, acts, rgbsig, emb, extra, want_emb, sigma_path_only=False):
    """The HIP backward of one NeRF evaluation over its dump: mf_nerf_backward_x (every pre-activation gradient +
    the embedded input's gradient) and ONE mf_weight_grads launch for all dW / db.
    -> ({parameter name: grad | None}, gpre, g_emb (P,64) | None).  ``sigma_path_only``: the call was
    NeRF(x, sigma_only=True) (nerf.py:81-95) -- xyz_encoding_final / extra_encoding / rgb did not take part and get
    no gradient, as under torch autograd."""
    D, W = m.D, m.W
    cin = m.in_channels_xyz
    names = [n for n, _ in m.named_parameters()]
    req = {n: p.requires_grad for n, p in m.named_parameters()}
    if sigma_path_only:
        for n in names:
            if n.startswith(("xyz_encoding_final", "extra_encoding", "rgb")):
                req[n] = False
    grads = {n: None for n in names}
    h = lambda l: acts[:, l * W:(l + 1) * W]
    f = acts[:, D * W:(D + 1) * W]
    gpre, ghead, g_emb_hip = nerf_backward_hip(m, g_out, acts, rgbsig, want_emb=want_emb)
    P, dev = acts.shape[0], acts.device
    gslot = lambda l: gpre[:, l * W:(l + 1) * W]
    g_e2 = gpre[:, (D + 1) * W:(D + 1) * W + W // 2]
    wants = lambda prefix: req[prefix + ".weight"] or req[prefix + ".bias"]
    emb64 = F.pad(emb, (0, 64 - emb.shape[1])) if emb.shape[1] < 64 else emb
    jobs, sinks = [], []

    def put(prefix, blocks, bias_from=0):
        """blocks: [(job index, row slice, col slice)] concatenated along the columns"""
        sinks.append((prefix, blocks, bias_from))

    for l in range(D):
        name = f"xyz_encoding_{l+1}.0"
        if not wants(name):
            continue
        blocks = []
        if l == 0 or l in m.skips:
            jobs.append((gslot(l), emb64, 256, 64, l == 0))
            blocks.append((len(jobs) - 1, slice(0, W), slice(0, cin)))
        if l > 0:
            jobs.append((gslot(l), h(l - 1), 256, 256, True))
            blocks.append((len(jobs) - 1, slice(0, W), slice(0, W)))
        put(name, blocks, blocks[-1][0])
    if wants("xyz_encoding_final"):
        jobs.append((gslot(D), h(D - 1), 256, 256, True))
        put("xyz_encoding_final", [(len(jobs) - 1, slice(0, W), slice(0, W))], len(jobs) - 1)
    if wants("extra_encoding.0"):
        jobs.append((g_e2, f, 128, 256, True))
        blocks = [(len(jobs) - 1, slice(0, W // 2), slice(0, W))]
        if extra is not None:
            ext32 = F.pad(extra, (0, 32 - extra.shape[1])) if extra.shape[1] < 32 else extra
            jobs.append((g_e2, ext32, 128, 32, False))
            blocks.append((len(jobs) - 1, slice(0, W // 2), slice(0, min(extra.shape[1], m.extra_feat_dim))))
        put("extra_encoding.0", blocks, blocks[0][0])
    head_job = None
    if wants("sigma") or wants("rgb.0"):
        jobs.append((ghead, acts[:, (D - 1) * W:(D - 1) * W + 640], 4, 640, True))
        head_job = len(jobs) - 1
    res = weight_grads(jobs, P, dev) if jobs else []
    for prefix, blocks, bias_from in sinks:
        if req[prefix + ".weight"]:
            parts = [res[j][0][rs, cs] for j, rs, cs in blocks]
            grads[prefix + ".weight"] = parts[0] if len(parts) == 1 else torch.cat(parts, 1)
        if req[prefix + ".bias"]:
            grads[prefix + ".bias"] = res[bias_from][1][blocks[0][1]]
    if head_job is not None:
        hW, hb = res[head_job]
        if req["sigma.weight"]:
            grads["sigma.weight"] = hW[3:4, 0:W]
        if req["sigma.bias"]:
            grads["sigma.bias"] = hb[3:4]
        if req["rgb.0.weight"]:
            grads["rgb.0.weight"] = hW[0:3, 2 * W:2 * W + W // 2]
        if req["rgb.0.bias"]:
            grads["rgb.0.bias"] = hb[0:3]
    return grads, gpre, g_emb_hip, emb64


class NerfModule(torch.autograd.Function):
    """``NeRF(inputs, sigma_only)`` called directly with gradients wanted (trainer_moco_flow.py:146-157 `forwarf_nerf`
    under the mask loss :337-362; nerf.py:61-102).  Forward: mf_nerf_forward_dump (the fused forward, also writing
    the per-sample layer outputs).  Backward: the same two launches as a render pass' NeRF node -- mf_nerf_backward_x
    (all pre-activation gradients + the gradient of the embedded input) and mf_weight_grads (every dW / db).
    A sigma_only call runs the full forward on a zero extra block (its rgb branch gets zero output gradient and its
    parameters None, as torch autograd leaves them)."""

    @staticmethod
    def forward(ctx, m, inputs, sigma_only, *params):
        x = inputs.detach().float()
        cin, ext_dim = m.in_channels_xyz, m.extra_feat_dim
        if sigma_only and ext_dim > 0:
            x = F.pad(x, (0, ext_dim))                        # the kernel reads the extra block of every row
        x = x.contiguous()
        B, dev = x.shape[0], x.device
        stride = (m.D + 1) * m.W + m.W // 2
        acts = torch.empty((B, stride), device=dev, dtype=torch.float32)
        out = torch.empty((B, 4), device=dev, dtype=torch.float32)
        desc, buf = m.packed()
        with torch.cuda.device(dev):
            L.check(L.lib().mf_nerf_forward_dump(desc, buf.data_ptr(), L.ptr(x), x.stride(0), B, L.ptr(out), L.ptr(acts),
                                                 stride, L.current_stream(dev)), "mf_nerf_forward_dump")
        ctx.m, ctx.sigma_only, ctx.in_grad, ctx.in_width = m, bool(sigma_only), inputs.requires_grad, inputs.shape[1]
        ctx.save_for_backward(acts, out, x)
        return out[:, 3:4].clone() if sigma_only else out.clone()

    @staticmethod
    def backward(ctx, g):
        m, sigma_only = ctx.m, ctx.sigma_only
        acts, rgbsig, x = ctx.saved_tensors
        D, W, cin, ext_dim = m.D, m.W, m.in_channels_xyz, m.extra_feat_dim
        names = [n for n, _ in m.named_parameters()]
        with torch.no_grad():
            if sigma_only:
                g_out = torch.zeros_like(rgbsig)
                g_out[:, 3] = g[:, 0]
            else:
                g_out = g.contiguous().float()
            emb = x[:, :cin]
            extra = x[:, cin:cin + ext_dim] if (ext_dim > 0 and not sigma_only) else None
            grads, gpre, g_emb, _ = nerf_fused_grads(m, g_out, acts, rgbsig, emb, extra, ctx.in_grad, sigma_path_only=sigma_only)
            g_in = None
            if ctx.in_grad:
                g_in = torch.zeros((x.shape[0], ctx.in_width), device=x.device, dtype=torch.float32)
                g_in[:, :cin] = g_emb[:, :cin]
                if extra is not None:                         # (B,128) x (128, extra_dim): the one library GEMM of this node
                    g_in[:, cin:] = gpre[:, (D + 1) * W:(D + 1) * W + W // 2] @ m.extra_encoding[0].weight[:, W:W + ext_dim]
        return (None, g_in, None) + tuple(grads[n] for n in names)


class NerfSamples(torch.autograd.Function):
    """Per-sample (rgb, sigma) of the canonical NeRF as an autograd node whose forward IS the fused HIP
    kernel's output (``rgbsig``, dumped).  Backward, per layer
        g_pre = g (.) (h > 0),   g_in = g_pre @ W,   dW = g_pre^T @ input,   db = sum g_pre:
    the g_pre / g_in chain of all layers is ONE fused HIP launch over the kernel's activation dump
    (mf_nerf_backward_x: the forward's register-resident MFMA core on the transposed weights, incl. the gradient of
    the embedded input); every dW / db comes from ONE persistent launch (mf_weight_grads) on (dump, g_pre).  No
    forward recompute, no autograd graph over the 12-layer MLP, no library GEMM.  Inputs that may need grad: the points
    ``xin`` (under NoF) and every NeRF parameter."""

    @staticmethod
    def forward(ctx, m, acts, rgbsig, emb_in, extra_in, emb_xyz, xin, sigma_path_only, *params):
        ctx.m, ctx.acts, ctx.emb_in, ctx.extra_in, ctx.emb_xyz = m, acts, emb_in, extra_in, emb_xyz
        # sigma_path_only: only sigma of this evaluation reaches the result (the coarse pass of a test_time render,
        # rendering.py:290-294): xyz_encoding_final / extra_encoding / rgb get no gradient, as under torch autograd
        ctx.sigma_path_only = bool(sigma_path_only)
        ctx.save_for_backward(rgbsig, xin)
        ctx.xin_grad = xin.requires_grad
        ctx.n_params = len(params)
        return rgbsig.detach()

    @staticmethod
    def backward(ctx, g_out):
        m, acts, emb, extra = ctx.m, ctx.acts, ctx.emb_in, ctx.extra_in
        if acts is None:
            raise RuntimeError("NerfSamples: backward ran twice through one forward (its activation dump is released after the first; "
                               "retain_graph is not supported by the explicit backward)")
        # the dump (9.7 KB per sample) is dead after this call: released HERE, not when the caller drops the result dict whose graph
        # holds this node -- a loop that keeps `results` until the next iteration overwrites it would otherwise hold two steps' dumps
        ctx.acts = ctx.emb_in = ctx.extra_in = None
        rgbsig, xin = ctx.saved_tensors
        D, W = m.D, m.W
        names = [n for n, _ in m.named_parameters()]
        req = {n: p.requires_grad for n, p in m.named_parameters()}
        grads = {n: None for n in names}
        cin = m.in_channels_xyz
        need_in = ctx.xin_grad

        with torch.no_grad():
            if ctx.sigma_path_only:
                g_out = g_out.clone()
                g_out[:, :3] = 0
            grads, gpre, g_emb_hip, emb64 = nerf_fused_grads(m, g_out, acts, rgbsig, emb, extra, need_in,
                                                             sigma_path_only=ctx.sigma_path_only)
            # the embedded input's gradient comes out of the chain launch itself; the sin / cos chain rule is one more
            # small launch (an embedding narrower than in_channels_xyz -- fewer frequencies -- reads its own columns only)
            g_xin = embed_backward_hip(ctx.emb_xyz, emb64, g_emb_hip) if need_in else None
        return (None, None, None, None, None, None, g_xin, None) + tuple(grads[n] for n in names)


def require_nerf_hip(m, P, under_nof):
    """The shapes the explicit NeRF backward is built for; anything else raises (there is no eager fallback)."""
    n_skip = len([s_ for s_ in m.skips if 0 < s_ < m.D])
    if not nerf_fused_eligible(m, P) or (under_nof and n_skip > 1):
        raise NotImplementedError(f"render_rays with gradients: the HIP backward is built for NeRFs with W = 256, in_channels_xyz <= "
                                  f"64, extra_feat_dim <= 32 (and at most one skip layer under NoF); got W={m.W}, D={m.D}, "
                                  f"skips={m.skips}, in_channels_xyz={m.in_channels_xyz}, extra_feat_dim={m.extra_feat_dim}")


class LossPartials(torch.autograd.Function):
    """The 12 (sum, count) loss partials of a training step as ONE node (the losses "fused
    into the epilogue"; models/losses.py:4-14, trainer/trainer_moco_flow.py:317-328).  Forward: mf_loss_partials on the
    arrays the fused passes already wrote (rgb, alphas, the per-sample consensus distances) -- the values of the
    gradient-free fast path, bit for bit.  Backward: mf_loss_partials_backward, one launch that writes the seeds
    2 g (rgb - gt) and  -g mask sign(x - recon) / 3  straight into the buffers CompositeSamples / NofPointsDumped
    consume.  No torch arithmetic, no mask tensor, no host sync.
    ``passes``: per pass a dict(planes = the kernel's output dict, z = depths (N,S)); differentiable inputs per pass:
    rgb (N,3), recon_local (N,S,3) | None, recon_global (N,S,3) | None."""

    @staticmethod
    def forward(ctx, partials_fn, rays, target, passes, *tensors):
        ctx.rays, ctx.target, ctx.passes = rays, target, passes
        ctx.shapes = [None if t is None else t.shape for t in tensors]
        out12 = partials_fn()
        ctx.save_for_backward(out12, *[t.detach() if t is not None else rays.new_empty(0) for t in tensors])
        return out12

    @staticmethod
    def backward(ctx, g12):
        out12, *tensors = ctx.saved_tensors
        rays, target, passes = ctx.rays, ctx.target, ctx.passes
        dev, N = rays.device, rays.shape[0]
        g12 = g12.detach().contiguous().double()
        descs, grads = [], []
        for q, ps in enumerate(passes):
            rgb, rl, rg = (tensors[3 * q + k] if ctx.shapes[3 * q + k] is not None else None for k in range(3))
            d = L.mf_loss_grad_pass()
            pl, z = ps["planes"], ps["z"]
            need = [ctx.needs_input_grad[4 + 3 * q + k] and ctx.shapes[3 * q + k] is not None for k in range(3)]
            g_rgb = torch.empty_like(rgb) if need[0] else None
            g_rl = torch.empty_like(rl) if need[1] else None
            g_rg = torch.empty_like(rg) if need[2] else None
            d.rgb, d.g_rgb = L.ptr(rgb.contiguous() if rgb is not None else None), L.ptr(g_rgb)
            if g_rl is not None or g_rg is not None:
                d.alphas, d.n_samples = L.ptr(pl["alphas"]), z.shape[1]
                d.rays, d.ray_stride, d.z_vals = L.ptr(rays), rays.stride(0), L.ptr(z)
                d.recon_local, d.g_recon_local = L.ptr(rl.contiguous() if g_rl is not None else None), L.ptr(g_rl)
                d.recon_global, d.g_recon_global = L.ptr(rg.contiguous() if g_rg is not None else None), L.ptr(g_rg)
            descs.append(d)
            grads += [g_rgb, g_rl, g_rg]
        while len(grads) < len(ctx.shapes):
            grads.append(None)
        with torch.cuda.device(dev):
            L.check(L.lib().mf_loss_partials_backward(C.byref(descs[0]), C.byref(descs[1]) if len(descs) > 1 else None,
                                                      L.ptr(target), N, out12.data_ptr(), g12.data_ptr(),
                                                      L.current_stream(dev)), "mf_loss_partials_backward")
        return (None, None, None, None) + tuple(grads)


class ConsensusMean(torch.autograd.Function):
    """``torch.mean`` of ONE consensus vector of a training pass -- what the unchanged trainer takes of every
    ``nof_*_disp_*`` entry right away (trainer/trainer_moco_flow.py:317-328) -- as one node instead of ~15 torch launches each
    way (|x - recon|, the mean over the three coordinates, the mask, two masked sums, their quotient, and the backward of all
    of them over (N, S, 3) tensors).  Forward: the mean mf_loss_partials takes over the per-sample distances the fused pass
    itself wrote (models/rendering.py:306-314; the two launches are shared by the vectors of the pass, lazy.ConsensusPass.stats).
    Backward: mf_loss_partials_backward with the one seed g / count -- ``-g mask sign(x - recon) / (3 count)`` straight into
    the buffer the NoF chain's last node consumes.  ``key``: "local" | "global"; ``planes``: the pass's output dict (alphas)."""

    @staticmethod
    def forward(ctx, group, key, planes, rays, z, recon):
        st = group.stats()
        ctx.key, ctx.alphas, ctx.rays, ctx.z, ctx.out12 = key, planes["alphas"], rays, z, st["out12"]
        ctx.save_for_backward(recon.detach())
        return st[key][2].clone()            # (the kernel's fp32 mean; a copy: the trainer adds into it in place)

    @staticmethod
    def backward(ctx, g):
        recon, = ctx.saved_tensors
        if g is None or not ctx.needs_input_grad[5]:
            return (None,) * 6
        rays, z, out12 = ctx.rays, ctx.z, ctx.out12
        dev, N = rays.device, rays.shape[0]
        slot = 4 if ctx.key == "local" else 8
        g12 = _onehot12(slot, dev) * (g.detach().double() / out12[slot + 1])
        recon = recon.contiguous()
        g_recon = torch.empty_like(recon)
        d = L.mf_loss_grad_pass()
        d.alphas, d.n_samples = L.ptr(ctx.alphas), z.shape[1]
        d.rays, d.ray_stride, d.z_vals = L.ptr(rays), rays.stride(0), L.ptr(z)
        if ctx.key == "local":
            d.recon_local, d.g_recon_local = L.ptr(recon), L.ptr(g_recon)
        else:
            d.recon_global, d.g_recon_global = L.ptr(recon), L.ptr(g_recon)
        with torch.cuda.device(dev):
            L.check(L.lib().mf_loss_partials_backward(C.byref(d), None, None, N, out12.data_ptr(), g12.data_ptr(),
                                                      L.current_stream(dev)), "mf_loss_partials_backward")
        return None, None, None, None, None, g_recon


_ONEHOT12 = {}


def _onehot12(slot, dev):
    key = (slot, str(dev))
    if key not in _ONEHOT12:
        v = torch.zeros(12, dtype=torch.float64, device=dev)
        v[slot] = 1.0
        _ONEHOT12[key] = v
    return _ONEHOT12[key]


class CompositeSamples(torch.autograd.Function):
    """(rgb, depth, opacity) of a pass as a function of the per-sample (rgb, sigma) plane.  Forward: the
    values the fused HIP pass already produced; backward: mf_composite_backward (rendering.py:157-192
    differentiated by hand: one wave per ray, product / suffix scans)."""

    @staticmethod
    def forward(ctx, rgbsig, rays, z_vals, noise, activation, background, rgb_val, depth_val, opacity_val):
        ctx.save_for_backward(rgbsig, rays, z_vals, noise if noise is not None else rgbsig.new_empty(0),
                              background if background is not None else rgbsig.new_empty(0))
        ctx.has_noise, ctx.has_bg, ctx.activation = noise is not None, background is not None, activation
        ctx.set_materialize_grads(False)                      # an output the loss does not use: a null seed, not zeros
        return rgb_val.detach(), depth_val.detach(), opacity_val.detach()

    @staticmethod
    def backward(ctx, g_rgb, g_depth, g_opacity):
        rgbsig, rays, z_vals, noise, background = ctx.saved_tensors
        N, S = z_vals.shape
        dev = rgbsig.device
        f = lambda t: None if t is None else t.contiguous().float()
        g_rgb, g_depth, g_opacity = f(g_rgb), f(g_depth), f(g_opacity)
        rays, z_vals, rgbsig = rays.contiguous(), z_vals.contiguous(), rgbsig.contiguous()
        noise = noise.contiguous() if ctx.has_noise else None
        background = background.contiguous() if ctx.has_bg else None
        g = torch.empty((N * S, 4), device=dev, dtype=torch.float32)
        ptr = lambda t: None if t is None else t.data_ptr()
        act = L.MF_ACT_RELU if ctx.activation == "relu" else L.MF_ACT_SOFTPLUS
        with torch.cuda.device(dev):
            L.check(L.lib().mf_composite_backward(rays.data_ptr(), rays.stride(0), N, S, z_vals.data_ptr(),
                                                  rgbsig.data_ptr(), ptr(noise), act, ptr(background), ptr(g_rgb),
                                                  ptr(g_depth), ptr(g_opacity), g.data_ptr(), L.current_stream(dev)),
                    "mf_composite_backward")
        return g, None, None, None, None, None, None, None, None
