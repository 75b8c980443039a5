"""Drop-in for /root/reference/models/nerf.py (class NeRF, :5-102)."""
import torch
from torch import nn

from . import _lib as L
from .packing import PackedWeights


class NeRF(nn.Module):
    """Canonical radiance field MLP. Same constructor, attributes, sub-module names
    and state_dict keys as the reference (xyz_encoding_{i}.0.*, xyz_encoding_final.*,
    extra_encoding.0.*, sigma.*, rgb.0.*); forward runs the fused HIP kernel."""

    def __init__(self, D=8, W=256, in_channels_xyz=33, skips=[4], extra_feat_type="none", extra_feat_dim=0):
        super().__init__()
        self.D = D
        self.W = W
        self.in_channels_xyz = in_channels_xyz
        self.skips = skips
        for i in range(D):
            if i == 0:
                layer = nn.Linear(in_channels_xyz, W)
            elif i in skips:
                layer = nn.Linear(W + in_channels_xyz, W)
            else:
                layer = nn.Linear(W, W)
            setattr(self, f"xyz_encoding_{i+1}", nn.Sequential(layer, nn.ReLU(True)))
        self.xyz_encoding_final = nn.Linear(W, W)
        self.extra_feat_type = extra_feat_type
        self.extra_feat_dim = extra_feat_dim
        assert extra_feat_type in ["none", "ind", "dir", "latent_code"], \
            f"extra_feat_type {extra_feat_type} for NeRF model not supported!!!"
        if extra_feat_type != "none":
            if extra_feat_type == "latent_code":
                self.app_code = torch.randn(1000, extra_feat_dim, requires_grad=True)
            self.extra_encoding = nn.Sequential(nn.Linear(W + extra_feat_dim, W // 2), nn.ReLU(True))
        else:
            self.extra_encoding = nn.Sequential(nn.Linear(W, W // 2), nn.ReLU(True))
        self.sigma = nn.Linear(W, 1)
        self.rgb = nn.Sequential(nn.Linear(W // 2, 3), nn.Sigmoid())
        self._packed = PackedWeights()
        self._packed_bf16 = PackedWeights()
        self._packed_x3 = PackedWeights()
        self._packed_bwd3 = PackedWeights()
        self._packed_bwd = PackedWeights()

    # ---- HIP plumbing -----------------------------------------------------
    def _build_desc(self):
        if self.extra_feat_type == "latent_code":
            raise NotImplementedError("NeRF model does not support latent code yet!!!")
        d = L.mf_nerf_desc()
        d.D, d.W, d.in_channels_xyz = self.D, self.W, self.in_channels_xyz
        mask = 0
        for s in self.skips:
            if 0 <= s < self.D:
                mask |= 1 << s
        d.skip_mask = mask
        d.extra_feat_type = {"none": L.MF_EXTRA_NONE, "ind": L.MF_EXTRA_IND, "dir": L.MF_EXTRA_DIR}[self.extra_feat_type]
        d.extra_feat_dim = self.extra_feat_dim
        keep = []

        def dp(t):
            t = t.detach().contiguous().float()
            keep.append(t)
            return t.data_ptr()

        if self.D + 1 > L.MF_MAX_LAYERS:
            raise NotImplementedError(f"NeRF with D={self.D} is not built (max {L.MF_MAX_LAYERS - 1})")
        for i in range(self.D):
            lin = getattr(self, f"xyz_encoding_{i+1}")[0]
            d.trunk_w[i], d.trunk_b[i] = dp(lin.weight), dp(lin.bias)
        d.final_w, d.final_b = dp(self.xyz_encoding_final.weight), dp(self.xyz_encoding_final.bias)
        d.extra_w, d.extra_b = dp(self.extra_encoding[0].weight), dp(self.extra_encoding[0].bias)
        d.sigma_w, d.sigma_b = dp(self.sigma.weight), dp(self.sigma.bias)
        d.rgb_w, d.rgb_b = dp(self.rgb[0].weight), dp(self.rgb[0].bias)
        return d, keep

    def packed(self, precision=L.MF_PREC_F32):
        """(descriptor, packed device buffer), re-packed when the parameters (or the precision) changed."""
        lib = L.lib()
        cache = {L.MF_PREC_F32: self._packed, L.MF_PREC_BF16: self._packed_bf16, L.MF_PREC_BF16X3: self._packed_x3}[precision]
        return cache.get(self, self._build_desc, lib.mf_nerf_packed_bytes_p, lib.mf_nerf_pack_p, "NeRF", precision)

    def invalidate_packed(self):
        """Drop the packed-weight caches (needed only after in-place edits through ``param.data``)."""
        for c in (self._packed, self._packed_bf16, self._packed_x3, self._packed_bwd, self._packed_bwd3):
            c.invalidate()

    def packed_bwd(self):
        """(descriptor, transposed fragment stream) for mf_nerf_backward; fp32 only."""
        lib = L.lib()
        return self._packed_bwd.get(self, self._build_desc, lambda d, _p: lib.mf_nerf_bwd_packed_bytes(d),
                                    lambda d, _p, buf, st: lib.mf_nerf_pack_bwd(d, buf, st), "NeRF backward", "bwd")

    def packed_bwd3(self):
        """(descriptor, transposed (hi, lo) fragment stream) for mf_nerf_backward3."""
        lib = L.lib()
        return self._packed_bwd3.get(self, self._build_desc, lambda d, _p: lib.mf_nerf_bwd3_packed_bytes(d),
                                     lambda d, _p, buf, st: lib.mf_nerf_pack_bwd3(d, buf, st), "NeRF backward (bf16x3)", "bwd3")

    def forward(self, inputs, sigma_only=False, img_ind=None):
        """inputs (B, in_channels_xyz [+ extra_feat_dim]) -> (B,4) rgb+sigma, or (B,1) sigma."""
        L.require_gpu(inputs, "NeRF.forward")
        if sigma_only:
            width = self.in_channels_xyz
        else:
            width = self.in_channels_xyz + self.extra_feat_dim   # torch.split contract, nerf.py:79
        if inputs.dim() != 2 or inputs.shape[1] != width:
            raise RuntimeError(f"NeRF.forward expects (B, {width}) [in_channels_xyz={self.in_channels_xyz}, "
                               f"extra_feat_dim={self.extra_feat_dim}, sigma_only={sigma_only}], "
                               f"got {tuple(inputs.shape)}")
        from . import autograd as A
        want_grad = A.needs_grad([self]) or (torch.is_grad_enabled() and inputs.requires_grad)
        if want_grad and inputs.shape[0] > 0:
            if not A.nerf_fused_eligible(self, inputs.shape[0]):
                raise NotImplementedError(f"NeRF.forward with gradients: the HIP backward is built for W = 256, one xyz block of <= 64 "
                                          f"columns and an extra block of <= 32 (got W={self.W}, in_channels_xyz={self.in_channels_xyz}, "
                                          f"extra_feat_dim={self.extra_feat_dim}, skips={self.skips}); there is no eager fallback")
            return A.NerfModule.apply(self, inputs, bool(sigma_only), *self.parameters())
        desc, buf = self.packed()
        x = inputs.detach().float()
        if x.stride(1) != 1:
            x = x.contiguous()
        B = x.shape[0]
        out = torch.empty((B, 1 if sigma_only else 4), device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            L.check(L.lib().mf_nerf_forward(desc, buf.data_ptr(), L.ptr(x), x.stride(0) if B else width, B,
                                            1 if sigma_only else 0, L.ptr(out), L.current_stream(x.device)),
                    "mf_nerf_forward")
        return out
