"""moco_flow_amd -- MI355X (gfx950) implementation of MoCo-Flow's volume-rendering hot path.

Drop-in surface (same names / signatures as /root/reference/models):
    Embedding, NeRF, NoF, get_model, get_loss, render_rays, sample_pdf
Everything computes in hand-written HIP kernels reached through the C ABI in
include/mocoflow_hip.h (libmocoflow_hip.so); there is no CPU or eager-PyTorch fallback.
"""
from torch import nn

from .embedding import Embedding
from .losses import MSELoss
from .nerf import NeRF
from .nof import NoF
from .points import query_sigma
from .rendering import render_rays, resample_merge, sample_pdf, set_precision, set_train_forward

__all__ = ["Embedding", "NeRF", "NoF", "get_model", "get_loss", "render_rays", "sample_pdf",
           "resample_merge", "set_precision", "set_train_forward", "query_sigma", "MSELoss"]


def get_model(model_config):
    """models/__init__.py:8-29 -- constructor arguments are passed positionally, in the
    reference's order."""
    if model_config['type'] == "Embedding":
        return Embedding(model_config['in_channels'], model_config['N_freqs'], model_config['logscale'])
    elif model_config['type'] == "NeRF":
        return NeRF(model_config['D'], model_config['W'], model_config['in_channels_xyz'],
                    model_config['skips'], model_config['extra_feat_type'], model_config['extra_feat_dim'])
    elif model_config['type'] == "NoF":
        return NoF(model_config['D'], model_config['W'], model_config['in_channels_xyz'],
                   model_config['skips'], model_config['extra_feat_type'], model_config['extra_feat_dim'],
                   model_config['use_quat'])
    else:
        raise ValueError('model type: {} not valid'.format(model_config['type']))


def get_loss(loss_config):
    """models/__init__.py:31-39."""
    if loss_config['type'] == "MSE":
        return MSELoss()
    elif loss_config['type'] == 'L1':
        return nn.L1Loss()
    elif loss_config['type'] == 'BCE':
        return nn.BCELoss()
    else:
        raise ValueError('loss type: {} not support'.format(loss_config['type']))
