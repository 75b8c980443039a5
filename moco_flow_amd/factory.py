"""Config-driven construction of the drop-in modules and losses.

Behavioural contract (checked in tests/test_host_cpu.py): same accepted ``type`` strings, same
positional constructor-argument order, same ValueError texts as the reference's factories
(models/__init__.py:8-29 for modules, :31-39 for losses).  The shape is a dispatch table: one
row per ``type`` = (constructor, the config keys handed over positionally, in that order).
"""
from __future__ import annotations

from typing import Callable, Dict, Mapping, Tuple

from torch import nn

from .embedding import Embedding
from .losses import MSELoss
from .nerf import NeRF
from .nof import NoF

_MLP_KEYS: Tuple[str, ...] = ("D", "W", "in_channels_xyz", "skips", "extra_feat_type", "extra_feat_dim")

MODEL_TABLE: Dict[str, Tuple[Callable[..., nn.Module], Tuple[str, ...]]] = {
    "Embedding": (Embedding, ("in_channels", "N_freqs", "logscale")),
    "NeRF": (NeRF, _MLP_KEYS),
    "NoF": (NoF, _MLP_KEYS + ("use_quat",)),
}

LOSS_TABLE: Dict[str, Callable[[], nn.Module]] = {
    "MSE": MSELoss,
    "L1": nn.L1Loss,
    "BCE": nn.BCELoss,
}


def get_model(model_config: Mapping) -> nn.Module:
    """Build Embedding / NeRF / NoF from a YAML ``model`` entry."""
    kind = model_config["type"]
    row = MODEL_TABLE.get(kind)
    if row is None:
        raise ValueError(f"model type: {kind} not valid")
    ctor, keys = row
    return ctor(*(model_config[k] for k in keys))


def get_loss(loss_config: Mapping) -> nn.Module:
    """Build the loss named by a YAML ``loss`` entry."""
    kind = loss_config["type"]
    ctor = LOSS_TABLE.get(kind)
    if ctor is None:
        raise ValueError(f"loss type: {kind} not support")
    return ctor()
