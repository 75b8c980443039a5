"""moco_flow_amd -- MI355X (gfx950) implementation of MoCo-Flow's volume-rendering hot path.

Drop-in surface (same names / signatures as the reference's ``models`` package):
    Embedding, NeRF, NoF, get_model, get_loss, render_rays, sample_pdf
Every forward value comes from hand-written HIP kernels reached through the C ABI in include/mocoflow_hip.h
(libmocoflow_hip.so); CPU tensors and a missing library raise.  The backward is HIP as well -- of render_rays passes
(which record gradients in fp32 whatever set_precision says) and of module-level NeRF / NoF / Embedding calls; shapes it
is not built for raise NotImplementedError under grad.  There is no eager fallback (autograd.py).
"""
from .embedding import Embedding
from .factory import get_loss, get_model
from .losses import MSELoss
from .nerf import NeRF
from .nof import NoF
from .points import query_sigma
from .autograd import set_dx_precision, set_wgrad_precision
from .rendering import render_rays, resample_merge, sample_pdf, set_precision, set_train_forward_precision

__all__ = ["Embedding", "NeRF", "NoF", "get_model", "get_loss", "render_rays", "sample_pdf",
           "resample_merge", "set_precision", "set_wgrad_precision", "set_dx_precision", "set_train_forward_precision", "query_sigma", "MSELoss"]
