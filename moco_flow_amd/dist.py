"""Ray-sharded multi-GPU rendering: one process per GPU, rays split into
contiguous blocks, no collective on the data path; the per-batch loss partial sums are
all-reduced (96 B; RCCL over xGMI on the GPU box: backend "nccl" IS RCCL on ROCm; gloo in CPU tests).

The reference itself never synchronises anything across ranks (its DDP wrapper is bypassed,
SURVEY.md §2c), so there is no reference multi-GPU numerics to match beyond "each rank renders
its own rays"; the contiguous split keeps the global ray order, so concatenating the ranks'
outputs reproduces the single-GPU result row for row.
"""
from __future__ import annotations

import contextlib
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
        slot and is reduced in place, without the staging copy (a 4 us launch on the step's stream).  Honoured only for
        gradient-free partials on the reducer's device (see below); otherwise the call takes the copy."""
        k = self.i % len(self.bufs)
        self.i += 1
        done = None
        if self.work[k] is not None:
            self.work[k].wait()                      # stream-ordered for NCCL/RCCL (no host block); blocking for gloo
            if collect:
                done = self.bufs[k].clone()
        # Donation is for GRADIENT-FREE partials only: under grad, `partials` is the output of autograd.LossPartials, which
        # saves it for its backward (the counts decide the all-true mask and the scaling) -- reducing it in place would
        # hand that backward the cross-rank sums (or trip autograd's version check).  A tensor on another device must not
        # replace the ring slot either.  Anything else takes the staging copy.
        if (donate and not partials.requires_grad and partials.grad_fn is None and partials.device == self.bufs[k].device
                and partials.dtype == self.bufs[k].dtype and partials.shape == self.bufs[k].shape and partials.is_contiguous()):
            self.bufs[k] = partials
        else:
            self.bufs[k].copy_(partials.detach())
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


_NEVER = object()           # member of a bucket's pending set that no hook ever removes


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


# ------------------------------------------------------------------ data-parallel TRAINING (training half)
class _GlobalSum(torch.autograd.Function):
    """y = all_reduce(SUM)(x) as a differentiable node.  Every rank then forms the SAME global loss L(y), and
    dL / dx_r = dL / dy on every rank (y = sum_r x_r), so the backward is the identity -- no collective in it.  With the
    loss formed from the globally reduced (sum, count) partials each rank's parameter gradients are its rays' share of
    the gradient of the GLOBAL loss: their SUM over the ranks (GradReducer, average=False) is exactly the single-process
    gradient on the concatenated batch, masked consensus means with data-dependent counts included."""

    @staticmethod
    def forward(ctx, x, group):
        y = x.detach().clone()
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(y, op=dist.ReduceOp.SUM, group=group)
        return y

    @staticmethod
    def backward(ctx, g):
        return g, None


def global_partials(partials: torch.Tensor, group=None) -> torch.Tensor:
    """The 12 loss partials of ``render_rays(..., _loss_target=gt)`` summed over the ranks (96 bytes through RCCL),
    differentiable: feed the result to ``losses.from_partials`` and call ``backward()``; pair with
    ``GradReducer(average=False)``."""
    return _GlobalSum.apply(partials, group)


class GradReducer:
    """Bucketed gradient all-reduce for ray-sharded data-parallel training, overlapped with the backward launches.

    The reference wraps its networks in DistributedDataParallel (trainer/base.py:251-256) but hands the BARE modules to
    render_rays (trainer_moco_flow.py:80-117, 203-206), so its reducer never fires and its "8-GPU training" is eight
    unsynchronised replicas.  This is what a drop-in that is meant to use eight GPUs for ONE model
    needs instead, sized for this path: 1.32 M fp32 gradients = 5.3 MB in total, four networks.

    * ONE flat fp32 buffer holds every gradient; bucket = one network (``buckets``: a list of modules, parameter lists or
      (name, module | parameters) pairs).  A parameter's ``.grad`` IS a view of its slot (DDP's gradient_as_bucket_view):
      autograd accumulates into it in place; after ``zero_grad(set_to_none=True)`` the first gradient of the step is
      copied in and ``.grad`` re-pointed at the slot.
    * ``register_post_accumulate_grad_hook`` on every trainable parameter: when the last parameter of a bucket that takes
      part in this step has its gradient, the bucket's slice goes out as ONE ``all_reduce(async_op=True)`` -- in the joint
      step the two NoFs' ``mf_weight_grads`` launches are enqueued before the NeRFs' dX chains finish, so their collectives
      run under the remaining backward launches.  xGMI is point to point (7 links x ~153 GB/s): a 2.4 MB NeRF bucket is
      ~16 us of link time per ring step, the 0.27 MB NoF buckets are latency; four collectives per step, never one per
      parameter.
    * ``wait()`` before ``optimizer.step()``: issues the buckets whose hooks did not complete (frozen sub-modules,
      parameters that got no gradient: every rank must issue the same collectives, and requires_grad is the same on
      every rank), waits for the work handles (stream-ordered for RCCL: no host block), and -- ``average=True`` (DDP's
      convention: per-rank mean losses) -- scales the whole flat buffer by 1 / world in ONE launch; ``average=False``
      with ``global_partials`` (the exact formulation, see there) needs no scaling at all.
    Parameters that got no gradient in a step keep ``.grad = None`` (the optimiser skips them, as in the reference's
    frozen-trunk phase, trainer_moco_flow.py:391-404); their slots travel as zeros."""

    def __init__(self, buckets, group=None, average: bool = True, device=None):
        self.group, self.average = group, average
        self.buckets = []
        params = []
        for i, b in enumerate(buckets):
            name = f"bucket{i}"
            if isinstance(b, tuple) and len(b) == 2 and isinstance(b[0], str):
                name, b = b
            ps = list(b.parameters()) if hasattr(b, "parameters") else list(b)
            if not ps:
                continue
            self.buckets.append({"name": name, "params": ps, "lo": 0, "hi": 0, "pending": None, "work": None, "ready": False,
                                 "index": len(self.buckets)})
            params += ps
        if not params:
            raise ValueError("GradReducer: no parameters")
        if any(p.dtype != torch.float32 for p in params):
            raise TypeError("GradReducer: fp32 parameters only (the path trains in fp32)")
        dev = device if device is not None else params[0].device
        off = 0
        self.slot = {}
        for b in self.buckets:
            b["lo"] = off
            for p in b["params"]:
                self.slot[id(p)] = (off, p.numel(), b)
                off += (p.numel() + 3) // 4 * 4                 # 16-byte aligned slots
            b["hi"] = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        self.active = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.active else 1
        self.issued = 0                                         # collectives issued so far (tests)
        self.issue_log = []                                     # bucket index of every collective, in issue order (tests)
        self._handles = {}
        self._defer = 0
        # issue order: reverse registration (the backward reaches the last network first) until the first wait() has agreed
        # on the order observed in the first backward
        self._order = list(range(len(self.buckets)))[::-1]
        self._order_agreed, self._observed = False, []
        self._register()

    def _register(self):
        # torch refuses hooks on tensors that do not require grad: a parameter frozen now gets its hook once it is unfrozen
        # (checked at every wait()).  Until it has one, its bucket is NOT issued from the hooks (see _hook: a gradient that
        # lands after the bucket went out would never be reduced) -- the whole bucket waits for wait(): costs overlap for one
        # step, never correctness (the coarse2fine transition unfreezes the NeRF trunk while rgb /
        # xyz_encoding_final / extra_encoding of the same bucket already train, trainer_moco_flow.py:391-404)
        for b in self.buckets:
            for p in b["params"]:
                if p.requires_grad and id(p) not in self._handles:
                    self._handles[id(p)] = p.register_post_accumulate_grad_hook(self._hook)

    def view_of(self, p):
        """A parameter's slot of the flat buffer, in the parameter's shape."""
        off, n, _ = self.slot[id(p)]
        return self.flat[off:off + n].view(p.shape)

    def _adopt(self, p):
        """p.grad -> the flat slot (first gradient of a step after zero_grad(set_to_none=True)); .grad becomes the view."""
        view = self.view_of(p)
        if p.grad.data_ptr() != view.data_ptr():
            view.copy_(p.grad)
            p.grad = view

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation over several backwards: inside this context the hooks only adopt the gradients into the
        flat buffer (autograd then accumulates in place), nothing is issued; run the LAST backward outside it (its hooks
        issue the buckets as usual) or call wait() right after the context (wait() issues what is left)."""
        self._defer += 1
        try:
            yield self
        finally:
            self._defer -= 1
            for b in self.buckets:
                b["pending"] = None

    def _hook(self, p):
        b = self.slot[id(p)][2]
        if b["work"] is not None:
            raise RuntimeError(f"GradReducer: gradient for bucket {b['name']} after its all-reduce was issued -- one backward per "
                               f"wait(); for gradient accumulation run all but the last backward under `with reducer.no_sync():`")
        if p.grad is not None:
            self._adopt(p)
        if self._defer:
            return
        if b["pending"] is None:
            b["pending"] = {id(q) for q in b["params"] if q.requires_grad and id(q) in self._handles}
            if any(q.requires_grad and id(q) not in self._handles for q in b["params"]):
                b["pending"].add(_NEVER)       # a trainable parameter without a hook (unfrozen since the last wait()): leave the bucket to wait()
        b["pending"].discard(id(p))
        if not b["pending"]:
            b["ready"] = True
            self._observed.append(b["index"])
            self._launch_ready()

    def _launch_ready(self):
        """Issue, IN self._order, every bucket that is ready and whose predecessors in that order have gone out: all ranks
        issue the same collectives in the same order on the communicator whatever order their hooks complete in (a rank
        with no rays, a pass whose NoF got no gradient there: ADVICE r5) -- what DDP does with its bucket indices."""
        for i in self._order:
            b = self.buckets[i]
            if b["work"] is not None:
                continue
            if not b["ready"]:
                return
            self._issue(b)

    def _issue(self, b):
        # slots of parameters without a gradient this step travel as zeros (their .grad stays None: the optimiser skips them)
        for p in b["params"]:
            if p.grad is None:
                self.view_of(p).zero_()
            else:
                self._adopt(p)
        seg = self.flat[b["lo"]:b["hi"]]
        if self.active:
            b["work"] = dist.all_reduce(seg, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            b["work"] = _Done()
        self.issued += 1
        self.issue_log.append(b["index"])

    def wait(self):
        """Before optimizer.step(): every bucket reduced, gradients averaged (average=True).  Returns the flat buffer."""
        for i in self._order:                                    # what the hooks did not complete: same fixed order
            if self.buckets[i]["work"] is None:
                self._issue(self.buckets[i])
        for b in self.buckets:
            b["work"].wait()
        if self.average and self.world > 1:
            self.flat.mul_(1.0 / self.world)
        if not self._order_agreed:
            # the order the buckets became ready in during the FIRST backward (rank 0's; the rest appended) becomes the issue
            # order of every later step on every rank: a fixed order that also matches the backward, so a bucket seldom waits
            # for a later one.  One 8-byte-per-bucket broadcast, once.
            seen = self._observed + [i for i in self._order if i not in self._observed]
            if self.active:
                t = torch.tensor(seen, dtype=torch.int64, device=self.flat.device)
                dist.broadcast(t, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
                seen = [int(x) for x in t.cpu().tolist()]
            self._order, self._order_agreed = seen, True
        self._observed = []
        for b in self.buckets:
            b["work"], b["pending"], b["ready"] = None, None, False
        self._register()
        return self.flat

    def remove(self):
        for h in self._handles.values():
            h.remove()
        self._handles = {}
