"""Drop-in for /root/reference/models/embedding.py (class Embedding, :4-47)."""
import torch
from torch import nn

from . import _lib as L


class Embedding(nn.Module):
    """x -> (x, w0 sin(f0 x), w0 cos(f0 x), w1 sin(f1 x), ...).

    Same constructor, attributes (N_freqs, in_channels, out_channels, freq_bands,
    weights, funcs) and ``set_weights`` contract as the reference; no parameters or
    buffers, hence no state_dict entries. ``forward`` runs mf_embedding_forward."""

    def __init__(self, in_channels, N_freqs, logscale=True):
        super().__init__()
        self.N_freqs = N_freqs
        self.in_channels = in_channels
        self.funcs = [torch.sin, torch.cos]
        self.out_channels = in_channels * (len(self.funcs) * N_freqs + 1)
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

    def descriptor(self) -> "L.mf_embedding":
        if self.N_freqs > L.MF_MAX_FREQS:
            raise NotImplementedError(f"Embedding with N_freqs={self.N_freqs} > {L.MF_MAX_FREQS} is not built")
        fb = self.freq_bands
        key = (getattr(fb, "_version", 0), tuple(self.weights))      # the trainer re-assigns / edits `weights`
        cached = self.__dict__.get("_desc_cache")
        # (`cached[2] is fb`: the cache keeps the tensor it was built from alive, so a re-assigned freq_bands can never
        #  be mistaken for it through a recycled id())
        if cached is not None and cached[2] is fb and cached[0] == key:
            return cached[1]
        d = L.mf_embedding()
        d.in_channels = self.in_channels
        d.n_freqs = self.N_freqs
        for k in range(self.N_freqs):
            d.freq[k] = float(fb[k])
            d.weight[k] = float(self.weights[k])
        self.__dict__["_desc_cache"] = (key, d, fb)
        return d

    def rows(self, x, repeat=1, width=None):
        """mf_embedding_forward_rows (no gradients): (B * repeat, width) -- the embedding of every row of x for
        ``repeat`` consecutive output rows, zero-padded to ``width`` columns; the X operands of the first-layer weight
        gradients (autograd.nerf_fused_grads) in one launch."""
        L.require_gpu(x, "Embedding.rows")
        if x.dim() != 2 or x.shape[1] != self.in_channels:
            raise RuntimeError(f"Embedding expects (B, {self.in_channels}), got {tuple(x.shape)}")
        width = self.out_channels if width is None else max(int(width), self.out_channels)
        xc = x.detach().contiguous().float()
        B = x.shape[0] * repeat
        out = torch.empty((B, width), device=x.device, dtype=torch.float32)
        with torch.cuda.device(x.device):
            L.check(L.lib().mf_embedding_forward_rows(self.descriptor(), L.ptr(xc), B, repeat, L.ptr(out), width,
                                                      L.current_stream(x.device)), "mf_embedding_forward_rows")
        return out

    def forward(self, x):
        L.require_gpu(x, "Embedding.forward")
        if x.dim() != 2 or x.shape[1] != self.in_channels:
            raise RuntimeError(f"Embedding expects (B, {self.in_channels}), got {tuple(x.shape)}")
        if torch.is_grad_enabled() and x.requires_grad:
            from . import autograd as A
            return A.EmbeddingModule.apply(self, x)
        xc = x.detach().contiguous().float()
        out = torch.empty((x.shape[0], self.out_channels), device=x.device, dtype=torch.float32)
        d = self.descriptor()
        with torch.cuda.device(x.device):
            L.check(L.lib().mf_embedding_forward(d, L.ptr(xc), x.shape[0], L.ptr(out),
                                                 L.current_stream(x.device)), "mf_embedding_forward")
        return out
