"""Timing-comparison switches (tools and tests only; NOT part of the drop-in surface): select the earlier
library-GEMM / torch-recompute variants of the training path so that the tools can time them against the shipped
HIP path on the same GPU."""
from moco_flow_amd import autograd as A
from moco_flow_amd import rendering


def _check(kind, allowed, what):
    if kind not in allowed:
        raise ValueError(f"{what}: {kind} not valid ({' | '.join(allowed)})")


def set_train_forward(mode: str):
    """"hip" (shipped): fused HIP forward + HIP backward nodes; "torch": the whole pass as eager device ops."""
    _check(mode, ("hip", "torch"), "train forward")
    rendering._TRAIN_FORWARD = mode


def set_nerf_backward(kind: str):
    _check(kind, ("hip", "gemm"), "nerf backward")
    A._NERF_BACKWARD = kind


def set_nof_backward(kind: str):
    _check(kind, ("hip", "torch"), "nof backward")
    A._NOF_BACKWARD = kind


def set_composite_backward(kind: str):
    _check(kind, ("hip", "torch"), "composite backward")
    A._COMPOSITE_BACKWARD = kind
