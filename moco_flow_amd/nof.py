"""Drop-in for /root/reference/models/nof.py (class NoF, :6-85)."""
import torch
from torch import nn

from . import _lib as L
from .packing import PackedWeights


class NoF(nn.Module):
    """Neural motion flow MLP (backward: observation -> canonical, forward: canonical ->
    observation). Same constructor, attributes and state_dict keys as the reference
    (nof_encoding_{i}.0.*, nof_encoding_final.*)."""

    def __init__(self, D=8, W=256, in_channels_xyz=33, skips=[4], extra_feat_type="ind", extra_feat_dim=0,
                 use_quat=False):
        super().__init__()
        self.D = D
        self.W = W
        self.in_channels_xyz = in_channels_xyz
        self.skips = skips
        self.use_quat = use_quat
        self.extra_feat_type = extra_feat_type
        self.extra_feat_dim = extra_feat_dim
        assert extra_feat_type in ["ind", "latent_code"], \
            f"extra_feat_type {extra_feat_type} for NoF model not supported!!!"
        if extra_feat_type == "latent_code":
            self.time_code = torch.randn(1000, extra_feat_dim, requires_grad=True)
        for i in range(D):
            if i == 0:
                layer = nn.Linear(in_channels_xyz + extra_feat_dim, W)
            elif i in skips:
                layer = nn.Linear(W + in_channels_xyz + extra_feat_dim, W)
            else:
                layer = nn.Linear(W, W)
            setattr(self, f"nof_encoding_{i+1}", nn.Sequential(layer, nn.ReLU(True)))
        self.nof_encoding_final = nn.Linear(W, 9 if use_quat else 3)
        self._packed = PackedWeights()
        self._packed_bf16 = PackedWeights()
        self._packed_x3 = PackedWeights()
        self._packed_bwd = PackedWeights()
        self._packed_bwd3 = PackedWeights()

    def _build_desc(self):
        if self.extra_feat_type == "latent_code":
            raise NotImplementedError("NoF model does not support latent code yet!!!")
        d = L.mf_nof_desc()
        d.D, d.W, d.in_channels_xyz, d.extra_feat_dim = self.D, self.W, self.in_channels_xyz, self.extra_feat_dim
        mask = 0
        for s in self.skips:
            if 0 <= s < self.D:
                mask |= 1 << s
        d.skip_mask = mask
        d.use_quat = 1 if self.use_quat else 0
        keep = []

        def dp(t):
            t = t.detach().contiguous().float()
            keep.append(t)
            return t.data_ptr()

        if self.D > L.MF_MAX_LAYERS:
            raise NotImplementedError(f"NoF with D={self.D} is not built")
        for i in range(self.D):
            lin = getattr(self, f"nof_encoding_{i+1}")[0]
            d.trunk_w[i], d.trunk_b[i] = dp(lin.weight), dp(lin.bias)
        d.head_w, d.head_b = dp(self.nof_encoding_final.weight), dp(self.nof_encoding_final.bias)
        return d, keep

    def invalidate_packed(self):
        """Drop the packed-weight caches (needed only after in-place edits through ``param.data``)."""
        for c in (self._packed, self._packed_bf16, self._packed_x3, self._packed_bwd, self._packed_bwd3):
            c.invalidate()

    def packed_bwd(self):
        """(descriptor, transposed fragment stream) for mf_nof_backward; fp32 only."""
        lib = L.lib()
        return self._packed_bwd.get(self, self._build_desc, lambda d, _p: lib.mf_nof_bwd_packed_bytes(d),
                                    lambda d, _p, buf, st: lib.mf_nof_pack_bwd(d, buf, st), "NoF backward", "bwd")

    def packed_bwd3(self):
        """(descriptor, transposed (hi, lo) bf16 fragment stream) for mf_nof_backward3."""
        lib = L.lib()
        return self._packed_bwd3.get(self, self._build_desc, lambda d, _p: lib.mf_nof_bwd3_packed_bytes(d),
                                     lambda d, _p, buf, st: lib.mf_nof_pack_bwd3(d, buf, st), "NoF backward (bf16x3)", "bwd3")

    def packed(self, precision=L.MF_PREC_F32):
        lib = L.lib()
        cache = {L.MF_PREC_F32: self._packed, L.MF_PREC_BF16: self._packed_bf16, L.MF_PREC_BF16X3: self._packed_x3}[precision]
        return cache.get(self, self._build_desc, lib.mf_nof_packed_bytes_p, lib.mf_nof_pack_p, "NoF", precision)

    def forward(self, inputs, xyz, img_ind=None):
        """inputs (B, in_channels_xyz + extra_feat_dim), xyz (B,3) -> (B,3)."""
        if self.extra_feat_type == "latent_code":
            raise NotImplementedError("NoF model does not support latent code yet!!!")
        L.require_gpu(inputs, "NoF.forward")
        width = self.in_channels_xyz + self.extra_feat_dim
        if inputs.dim() != 2 or inputs.shape[1] != width or xyz.shape != (inputs.shape[0], 3):
            raise RuntimeError(f"NoF expects inputs (B, {width}) and xyz (B, 3), got {tuple(inputs.shape)}, {tuple(xyz.shape)}")
        from . import autograd as A
        wrt = [t for t in (inputs, xyz) if torch.is_grad_enabled() and t.requires_grad]
        B = inputs.shape[0]
        if (A.needs_grad([self]) or wrt) and B > 0:
            # training call on data points (stage 2, SMPL-point losses: trainer_nof.py:85-112, trainer_moco_flow.py:159-187,
            # 337-362 -- the points are data, only the parameters learn): HIP forward-with-dump + HIP backward
            if wrt:
                raise NotImplementedError("NoF.forward: gradients w.r.t. `inputs` / `xyz` of a module-level call are not built (the "
                                          "reference's trainers call it on data points; render_rays differentiates its chains "
                                          "through the points itself); there is no eager fallback")
            if not A.nof_hip_supported(self, None):
                raise NotImplementedError(f"NoF.forward with gradients: the HIP backward is built for W = 128, in_channels_xyz = 33, "
                                          f"extra_feat_dim = 33, at most one skip layer (got W={self.W}, D={self.D}, skips={self.skips})")
            return A.NofModule.apply(self, inputs, xyz, *self.parameters())
        desc, buf = self.packed()
        x = inputs.detach().float()
        if x.stride(1) != 1:
            x = x.contiguous()
        p = xyz.detach().float().contiguous()
        out = torch.empty((B, 3), device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            L.check(L.lib().mf_nof_forward(desc, buf.data_ptr(), L.ptr(x), x.stride(0) if B else width, L.ptr(p), B,
                                           L.ptr(out), L.current_stream(x.device)), "mf_nof_forward")
        return out
