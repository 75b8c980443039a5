"""Ray-sharded multi-GPU rendering: one process per GPU, rays split into
contiguous blocks, no collective on the data path; the per-batch loss partial sums are
all-reduced (RCCL over xGMI on the GPU box: backend "nccl" IS RCCL on ROCm; gloo in CPU tests).

The reference itself never synchronises anything across ranks (its DDP wrapper is bypassed,
SURVEY.md §2c), so there is no reference multi-GPU numerics to match beyond "each rank renders
its own rays"; the contiguous split keeps the global ray order, so concatenating the ranks'
outputs reproduces the single-GPU result row for row.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n_rays: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of rank `rank`; the first n_rays % world ranks get one more ray."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError(f"bad rank/world {rank}/{world}")
    base, rem = divmod(n_rays, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def loss_partials(result: Dict[str, torch.Tensor], target: torch.Tensor) -> torch.Tensor:
    """[sum (rgb_c-gt)^2, n_c, sum (rgb_f-gt)^2, n_f, sum nof_local, n_local, sum nof_global, n_global]
    in float64: the additive pieces of MSELoss (models/losses.py:4-14) and of the consensus
    means (trainer_moco_flow.py:317-328) over this rank's rays."""
    p = torch.zeros(8, dtype=torch.float64, device=target.device)
    if "rgb_coarse" in result:
        d = (result["rgb_coarse"] - target).double()
        p[0], p[1] = (d * d).sum(), d.numel()
    if "rgb_fine" in result:
        d = (result["rgb_fine"] - target).double()
        p[2], p[3] = (d * d).sum(), d.numel()
    for i, key in ((4, "nof_local_disp"), (6, "nof_global_disp")):
        s = n = 0.0
        for tag in ("coarse", "fine"):
            v = result.get(f"{key}_{tag}")
            if v is not None:
                s = s + v.double().sum()
                n = n + v.numel()
        p[i], p[i + 1] = s, n
    return p


def reduce_loss(partials: torch.Tensor, group=None) -> Dict[str, float]:
    """All-reduce(SUM) the 64-byte partial vector and turn it into the global means."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(partials, op=dist.ReduceOp.SUM, group=group)
    p = partials.tolist()
    mean = lambda s, n: s / n if n > 0 else 0.0
    return {"mse_coarse": mean(p[0], p[1]), "mse_fine": mean(p[2], p[3]),
            "nof_local": mean(p[4], p[5]), "nof_global": mean(p[6], p[7]),
            "img_loss": mean(p[0], p[1]) + mean(p[2], p[3])}


class OverlappedLossReducer:
    """All-reduce of the per-step loss partials that does not serialise with the next step's kernels.

    A blocking ``all_reduce`` makes the launch stream wait for the collective after every step (tens of
    microseconds of xGMI latency against a 2.3 ms render pass).  Here the collective of step i is issued
    asynchronously on one of ``depth`` rotating buffers and only waited for when its buffer comes round
    again (``depth`` steps later) or at ``finish()``; step i+1's kernels are enqueued right behind step i's.
    ``push(..., collect=True)`` returns the finished totals of the step that previously used the buffer."""

    def __init__(self, n: int, device, depth: int = 2, group=None, dtype=torch.float64):
        self.bufs = [torch.zeros(n, dtype=dtype, device=device) for _ in range(depth)]
        self.work = [None] * depth
        self.group = group
        self.i = 0
        self.active = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1

    def push(self, partials: torch.Tensor, collect: bool = False) -> Optional[torch.Tensor]:
        k = self.i % len(self.bufs)
        self.i += 1
        done = None
        if self.work[k] is not None:
            self.work[k].wait()                      # stream-ordered for NCCL/RCCL (no host block); blocking for gloo
            if collect:
                done = self.bufs[k].clone()
        self.bufs[k].copy_(partials)
        if self.active:
            self.work[k] = dist.all_reduce(self.bufs[k], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            self.work[k] = _Done()
        return done

    def finish(self):
        """Wait for everything in flight; returns the totals in issue order of the still-pending steps."""
        out = []
        n = len(self.bufs)
        for j in range(n):
            k = (self.i + j) % n
            if self.work[k] is not None:
                self.work[k].wait()
                out.append(self.bufs[k].clone())
                self.work[k] = None
        return out


class _Done:
    def wait(self):
        return True


def render_sharded(render: Callable[..., Dict[str, torch.Tensor]], rays: torch.Tensor,
                   background: Optional[torch.Tensor], *args, rank: Optional[int] = None,
                   world: Optional[int] = None, **kwargs) -> Tuple[Dict[str, torch.Tensor], Tuple[int, int]]:
    """Render this rank's contiguous block of `rays` with `render` (render_rays signature)."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    lo, hi = shard_bounds(rays.shape[0], rank, world)
    out = render(rays[lo:hi], None if background is None else background[lo:hi], *args, **kwargs)
    return out, (lo, hi)


def gather_pixels(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """All-gather per-ray outputs (N_local, ...) of unequal block sizes back into global ray order."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    sizes = [shard_bounds(n_total, r, world) for r in range(world)]
    mx = max(h - l for l, h in sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([parts[r][: sizes[r][1] - sizes[r][0]] for r in range(world)], 0)
