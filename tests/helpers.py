"""Shared builders: turn a golden render case into (rays, bg, embeddings, models, kwargs)
for either backend -- ``oracle.cpu_ref`` (CPU checker) or ``moco_flow_amd`` (HIP product)."""
import os

import numpy as np
import torch

from cases import RENDER_CASES  # tests/golden/cases.py
from moco_flow_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def relerr(a, b):
    """max|a-b| / max|b|  (SURVEY.md §8d 'max-rel')."""
    a = torch.as_tensor(a).detach().to(torch.float64).cpu()
    b = torch.as_tensor(b).detach().to(torch.float64).cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    if b.numel() == 0:
        return 0.0
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def build_case(backend, case, seed, device="cpu", tags=None):
    """backend: a module exposing Embedding / NeRF / NoF classes with the reference's
    constructor signatures and ``load_state_dict``."""
    c = case if isinstance(case, dict) else RENDER_CASES[case]
    tags = tags or {}          # weight-stream names per role (bench.py draws "nerf" / "nerf_fine" / "bw" / "fw")
    nof = c.get("nof", "none")
    extra = c["extra"]
    extra_dim = {"dir": 27, "ind": 5, "none": 0}[extra]

    def emb(cin, nf, weights=None):
        e = backend.Embedding(cin, nf, True)
        if weights is not None:
            e.weights = list(weights)
        return e

    def nerf(tag):
        m = backend.NeRF(8, 256, 63, [4], extra, extra_dim)
        sd = synth.nerf_state(seed, extra_feat_type=extra, extra_feat_dim=extra_dim,
                              regime=c["regime"], tag=tags.get(tag, tag))
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        return m.to(device) if hasattr(m, "to") else m

    def nof_model(tag):
        m = backend.NoF(4, 128, 33, [2], "ind", 33, c.get("quat", True))
        sd = synth.nof_state(seed, use_quat=c.get("quat", True), tag=tags.get(tag, tag), head_scale=0.25)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        return m.to(device) if hasattr(m, "to") else m

    nerf_embs = [emb(3, c.get("xyz_freqs", 10), c.get("xyz_w")),
                 emb(1, 2) if extra == "ind" else None,
                 emb(3, 4) if extra == "dir" else None]
    nerfs = [nerf("coarse")] + ([nerf("fine")] if c["M"] > 0 else [])
    nof_embs = nof_models = None
    if nof != "none":
        nof_embs = [emb(3, 5), emb(1, 16)]
        nof_models = [nof_model("bw")] + ([nof_model("fw")] if nof in ("local", "global") else [])
    kw = dict(nof_embeddings=nof_embs, nof_models=nof_models,
              chain_local=nof in ("local", "global"), chain_global=nof == "global",
              N_samples=c["S"], N_importance=c["M"], use_disp=c.get("disp", False), perturb=0,
              noise_std=0, nerf_activate_type=c.get("act", "relu"), test_time=c.get("test", False))
    return nerf_embs, nerfs, kw


def case_inputs(c, seed, n=None, device="cpu"):
    nof = c.get("nof", "none")
    n = c["n"] if n is None else n
    rays, bg = synth.rays(seed, n, chained=(nof == "global"))
    rays = torch.from_numpy(rays).to(device)
    bg = torch.from_numpy(bg).to(device) if c.get("bg", True) else None
    return rays, bg


class OracleOps:
    """CPU stand-ins, backed by oracle.cpu_ref, for the building blocks the backward unit tests differentiate: the same
    weights as a product module in an oracle container whose tensors are autograd leaves (``twin``), and the oracle's
    embedding / network / chain / composite functions on CPU copies of the inputs.  (Round 2 compared the HIP nodes with
    an eager restatement that lived inside the package; the checker is the oracle.)"""

    def __init__(self, R):
        self.R = R
        self.twins = {}

    def twin(self, m):
        if id(m) not in self.twins:
            sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
            if hasattr(m, "use_quat"):
                t = self.R.NoF(m.D, m.W, m.in_channels_xyz, list(m.skips), m.extra_feat_type, m.extra_feat_dim, m.use_quat, state=sd)
            else:
                t = self.R.NeRF(m.D, m.W, m.in_channels_xyz, list(m.skips), m.extra_feat_type, m.extra_feat_dim, state=sd)
            for k in t.p:
                t.p[k] = t.p[k].clone().requires_grad_(True)
            self.twins[id(m)] = t
        return self.twins[id(m)]

    def grads(self, m):
        return {k: v.grad for k, v in self.twin(m).p.items()}

    def emb(self, e):
        o = self.R.Embedding(e.in_channels, e.N_freqs)
        o.freq_bands = torch.as_tensor(e.freq_bands).detach().clone().cpu().float()
        o.weights = [float(w) for w in e.weights]
        return o

    @staticmethod
    def _c(t):
        return t if (t is None or not torch.is_tensor(t) or not t.is_cuda) else t.detach().cpu()

    def embed(self, e, x):
        return self.emb(e)(x if not x.is_cuda else self._c(x))

    def pad_to(self, t, width):
        out = torch.zeros((t.shape[0], width))
        out[:, :t.shape[1]] = t
        return out

    def nerf_forward(self, m, inputs, sigma_only=False):
        return self.twin(m)(inputs, sigma_only=sigma_only)

    def nof_forward(self, m, inputs, xyz):
        return self.twin(m)(inputs, xyz)

    def nof_points(self, xyz, ind, embs, m):
        return self.R.nof_inference(xyz, self._c(ind), [self.emb(embs[0]), self.emb(embs[1])], self.twin(m))

    def composite(self, rgbsig, z_vals, rays_d, noise, activation, background):
        N, S = z_vals.shape
        rs = rgbsig.view(N, S, 4)
        nz = torch.zeros(N, S) if noise is None else self._c(noise)
        rgb, depth, weights, alphas = self.R.composite(rs[..., 3], rs[..., :3], self._c(z_vals), self._c(rays_d), nz, activation,
                                                       self._c(background))
        return dict(rgb=rgb, depth=depth, opacity=weights.sum(1), weights=weights, alphas=alphas)


def pad_to(t, width):
    """(P, w) -> (P, width) zero padded on the right (the narrow operands the backward nodes also accept)."""
    if t.shape[1] == width:
        return t
    out = t.new_zeros((t.shape[0], width))
    out[:, :t.shape[1]] = t
    return out
