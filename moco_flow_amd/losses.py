"""Photometric loss over the result dict of ``render_rays`` (reference: models/losses.py:4-26).

The value is mean((rgb_coarse - target)^2) plus, when a fine pass ran, the same term on
``rgb_fine`` -- two reductions over 3 floats per ray, plain device ops and not a kernel of the hot
path.  The sharded variant that keeps the global mean exact under ragged ray
shards lives in ``dist.loss_partials`` / ``dist.reduce_loss``.
"""
from __future__ import annotations

from typing import Mapping

import torch
import torch.nn.functional as F
from torch import nn

_PASSES = ("coarse", "fine")


class MSELoss(nn.Module):
    """Sum over the rendered passes of the mean squared colour error."""

    def forward(self, inputs: Mapping[str, torch.Tensor], targets: torch.Tensor) -> torch.Tensor:
        terms = [F.mse_loss(inputs[f"rgb_{p}"], targets, reduction="mean")
                 for p in _PASSES if f"rgb_{p}" in inputs]
        if len(terms) == 0 or "rgb_coarse" not in inputs:
            raise KeyError("rgb_coarse")
        total = terms[0]
        for t in terms[1:]:
            total = total + t
        return total


def from_partials(partials: torch.Tensor):
    """The reference's loss terms from the 12 (sum, count) partials of ``render_rays(..., _loss_target=gt)``
    (mf_loss_partials; layout of dist.loss_partials), as differentiable device scalars with no host sync:
    ``img_loss`` = MSE coarse + fine (models/losses.py:4-14), ``nof_local`` / ``nof_global`` = mean over the
    masked points, coarse + fine (trainer_moco_flow.py:317-328).  A term whose count is 0 (pass / chain absent) is 0."""
    img, local, glob = _PartialMeans.apply(partials).unbind(0)
    return {"img_loss": img, "nof_local": local, "nof_global": glob}


class _PartialMeans(torch.autograd.Function):
    """(12,) partials [sum, count] x 6 -> (3,) [img, local, global], each the coarse + fine mean (0 where the count is 0):
    a handful of vectorised device ops forward and backward (d term / d sum = 1 / count; the counts carry no gradient)
    instead of ~100 scalar ones through autograd."""

    @staticmethod
    def forward(ctx, partials):
        p = partials.detach().view(6, 2)
        den = torch.where(p[:, 1] > 0, p[:, 1], torch.ones_like(p[:, 1]))
        ctx.save_for_backward(den)
        return (p[:, 0] / den).view(3, 2).sum(1)

    @staticmethod
    def backward(ctx, g):
        den, = ctx.saved_tensors
        g6 = g.to(den.dtype).repeat_interleave(2) / den
        return torch.stack((g6, torch.zeros_like(g6)), 1).view(12)
