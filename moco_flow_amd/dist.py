"""Ray-sharded multi-GPU rendering: one process per GPU, rays split into
contiguous blocks, no collective on the data path; the per-batch loss partial sums are
all-reduced (96 B; RCCL over xGMI on the GPU box: backend "nccl" IS RCCL on ROCm; gloo in CPU tests).

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


N_PARTIALS = 12     # 6 (sum, count) pairs: mse_c, mse_f, local_c, local_f, global_c, global_f


def loss_partials(result: Dict[str, torch.Tensor], target: torch.Tensor) -> torch.Tensor:
    """The additive pieces of the training losses over this rank's rays, float64, one (sum, count) pair per
    term the reference averages SEPARATELY:
      [0:4]  sum (rgb_coarse-gt)^2, n | sum (rgb_fine-gt)^2, n            MSELoss, models/losses.py:4-14
      [4:8]  sum nof_local_disp_coarse, n | sum nof_local_disp_fine, n     trainer_moco_flow.py:317-321
      [8:12] sum nof_global_disp_coarse, n | sum nof_global_disp_fine, n   trainer_moco_flow.py:323-327
    (the reference adds mean(coarse) + mean(fine); pooling both passes into one mean would halve the term
    and weight it by the mask counts).  Built with one stack: no per-slot indexed writes on the device."""
    dev = target.device
    zero = torch.zeros((), dtype=torch.float64, device=dev)
    parts = []
    for key in ("rgb_coarse", "rgb_fine"):
        v = result.get(key)
        if v is None:
            parts += [zero, zero]
        else:
            d = (v - target).double()
            parts += [(d * d).sum(), torch.full((), float(d.numel()), dtype=torch.float64, device=dev)]
    for key in ("nof_local_disp", "nof_global_disp"):
        for tag in ("coarse", "fine"):
            v = result.get(f"{key}_{tag}")
            if v is None:
                parts += [zero, zero]
            else:
                parts += [v.double().sum(), torch.full((), float(v.numel()), dtype=torch.float64, device=dev)]
    return torch.stack(parts)


def reduce_loss(partials: torch.Tensor, group=None) -> Dict[str, float]:
    """All-reduce(SUM) the 96-byte partial vector and turn it into the reference's global loss terms:
    img_loss = mse_coarse + mse_fine, nof_local = mean_coarse + mean_fine, nof_global likewise."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(partials, op=dist.ReduceOp.SUM, group=group)
    p = partials.tolist()
    mean = lambda i: p[i] / p[i + 1] if p[i + 1] > 0 else 0.0
    out = {"mse_coarse": mean(0), "mse_fine": mean(2),
           "nof_local_coarse": mean(4), "nof_local_fine": mean(6),
           "nof_global_coarse": mean(8), "nof_global_fine": mean(10)}
    out["img_loss"] = out["mse_coarse"] + out["mse_fine"]
    out["nof_local"] = out["nof_local_coarse"] + out["nof_local_fine"]
    out["nof_global"] = out["nof_global_coarse"] + out["nof_global_fine"]
    return out


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
        # an initialised process group is used at ANY world size (a 1-rank group still issues the collective through
        # RCCL: tests/rccl_child.py runs exactly that on one GPU); without a group the pushes are plain copies
        self.active = dist.is_available() and dist.is_initialized()

    def push(self, partials: torch.Tensor, collect: bool = False, donate: bool = False) -> Optional[torch.Tensor]:
        """``donate``: the caller gives ``partials`` up (render_rays returns a fresh tensor every step) -- it becomes the ring
        slot and is reduced in place, without the staging copy (a 4 us launch on the step's stream)."""
        k = self.i % len(self.bufs)
        self.i += 1
        done = None
        if self.work[k] is not None:
            self.work[k].wait()                      # stream-ordered for NCCL/RCCL (no host block); blocking for gloo
            if collect:
                done = self.bufs[k].clone()
        if donate and partials.dtype == self.bufs[k].dtype and partials.shape == self.bufs[k].shape and partials.is_contiguous():
            self.bufs[k] = partials
        else:
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
