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
    p = partials
    one = torch.ones((), dtype=p.dtype, device=p.device)

    def mean(i):
        return p[i] / torch.where(p[i + 1] > 0, p[i + 1], one)

    return {"img_loss": mean(0) + mean(2), "nof_local": mean(4) + mean(6), "nof_global": mean(8) + mean(10)}
