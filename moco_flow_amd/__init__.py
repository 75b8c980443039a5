"""moco_flow_amd -- MI355X (gfx950) implementation of MoCo-Flow's volume-rendering hot path.

Drop-in surface (same names / signatures as the reference's ``models`` package):
    Embedding, NeRF, NoF, get_model, get_loss, render_rays, sample_pdf
Everything computes in hand-written HIP kernels reached through the C ABI in
include/mocoflow_hip.h (libmocoflow_hip.so); there is no CPU or eager-PyTorch fallback.
"""
from .embedding import Embedding
from .factory import get_loss, get_model
from .losses import MSELoss
from .nerf import NeRF
from .nof import NoF
from .points import query_sigma
from .rendering import render_rays, resample_merge, sample_pdf, set_precision, set_train_forward

__all__ = ["Embedding", "NeRF", "NoF", "get_model", "get_loss", "render_rays", "sample_pdf",
           "resample_merge", "set_precision", "set_train_forward", "query_sigma", "MSELoss"]
