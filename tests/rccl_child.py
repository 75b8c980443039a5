"""The collective leg of BASELINE config C4 as far as ONE GPU can execute it (run as a fresh process).

Reference launch model: one process per GPU, `torch.distributed.init_process_group(backend="nccl")`
(trainer/base.py:104-106, README.md:127-145).  Here: a 1-rank "nccl" (= RCCL on ROCm) group on cuda:0, then 50
steps of the MoCo bf16 pass whose 12 loss partials (mf_loss_partials) go through
`dist.OverlappedLossReducer.push` -- real `ncclAllReduce` Work objects on rotating buffers, waited for two steps
later.  Checks: (i) every step's reduced totals equal its un-reduced partials bit for bit (SUM over one rank);
(ii) the pushes raise nothing under `torch.cuda.set_sync_debug_mode("error")` (no host synchronisation: the
step stays launch-only); (iii) `reduce_loss` on the reduced vector gives the reference's loss terms of the
un-reduced one; (iv) round 5: ten joint training steps whose real flat gradient goes through `dist.GradReducer`
(four buckets, ncclAllReduce from the post-accumulate hooks, global loss from `dist.global_partials`) under the same
sync-debug mode, bit-identical to the un-reduced gradients.  Prints one JSON line; exit code 0 = pass.

Started by tests/conftest.py at session start -- BEFORE the pytest process touches the GPU -- because a process
that has initialised the GPU must not exec another program on this pool; `test_rccl_one_rank_child` asserts on
its output."""
import json
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def main():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend="nccl", world_size=1, rank=0, device_id=dev)
    out = {"backend": dist.get_backend(), "world": dist.get_world_size()}

    import moco_flow_amd as M
    from moco_flow_amd import losses, rendering, synth
    from moco_flow_amd.dist import N_PARTIALS, OverlappedLossReducer, reduce_loss
    from cases import RENDER_CASES
    from helpers import build_case
    M._lib.lib()
    rendering.STRICT_RNG = False
    n, steps = 512, 50
    embs, nerfs, kw = build_case(M, dict(RENDER_CASES["r_moco_global"]), 0, device="cuda")
    rays_np, bg_np = synth.rays(0, n, chained=True)
    rays, bg = torch.from_numpy(rays_np).to(dev), torch.from_numpy(bg_np).to(dev)
    gts = [torch.rand(n, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(i)) for i in range(steps)]
    rendering.set_precision("bf16")
    red = OverlappedLossReducer(N_PARTIALS, dev, depth=2)
    assert red.active, "a 1-rank process group must still route the partials through the collective"
    local, reduced = [], []

    def step(i):
        res = M.render_rays(rays, bg, embs, nerfs, _loss_target=gts[i], **kw)
        local.append(res["loss_partials"].clone())
        # odd steps donate the kernel's own tensor: it becomes the ring slot and ncclAllReduce runs in place on it (what bench.py does)
        done = red.push(res["loss_partials"], collect=True, donate=i % 2 == 1)
        if done is not None:
            reduced.append(done)

    with torch.no_grad():
        for i in range(3):                       # communicator / kernel warm-up outside the sync-debug region
            step(i)
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("error")
        try:
            for i in range(3, steps):
                step(i)
        finally:
            torch.cuda.set_sync_debug_mode("default")
        work_types = sorted({type(w).__name__ for w in red.work if w is not None})
        reduced += red.finish()
    torch.cuda.synchronize()
    out["work_types"] = work_types
    out["steps"] = len(reduced)
    assert len(reduced) == steps, (len(reduced), steps)
    assert all("Done" not in t for t in work_types), work_types      # real c10d Work objects, not the no-group stub
    mism = sum(0 if torch.equal(a, b) else 1 for a, b in zip(local, reduced))
    out["mismatching_steps"] = mism
    assert mism == 0
    # distinct targets -> distinct partials: the rotation did not hand a buffer back early
    assert len({float(p[0]) for p in local}) == steps
    t_red = reduce_loss(reduced[-1].clone())
    t_loc = losses.from_partials(local[-1])
    for k in ("img_loss", "nof_local", "nof_global"):
        assert abs(t_red[k] - float(t_loc[k])) <= 1e-12 * max(1.0, abs(t_red[k])), (k, t_red[k], float(t_loc[k]))
    out["img_loss"] = t_red["img_loss"]

    # ---- the gradient leg (training half): 10 joint MoCo training steps whose REAL flat gradient (two NeRF-sized
    # and two NoF buckets) goes through dist.GradReducer's ncclAllReduce Work objects, issued from the post-accumulate
    # hooks inside backward(), with the global loss formed from the all-reduced partials (dist.global_partials) -- under
    # sync-debug "error": the data-parallel step stays launch-only.  World 1: SUM is the identity, so the reduced flat
    # buffer must equal the gradients of the same step without a reducer bit for bit (the kernels are deterministic).
    from moco_flow_amd.dist import GradReducer, global_partials
    rendering.set_precision("f32")
    embs, nerfs, kw = build_case(M, dict(RENDER_CASES["r_moco_global_fine"]), 0, device="cuda")     # coarse + fine NeRF, bw + fw NoF
    nets = list(nerfs) + list(kw["nof_models"])
    gsteps = 10

    def total_of(parts):
        t = losses.from_partials(parts)
        return t["img_loss"] + 0.1 * t["nof_local"] + 0.1 * t["nof_global"]

    def train_step(i, red):
        for m in nets:
            m.zero_grad(set_to_none=True)
        res = M.render_rays(rays[:256], bg[:256], embs, nerfs, _loss_target=gts[i][:256], **kw)
        parts = res["loss_partials"] if red is None else global_partials(res["loss_partials"])
        total_of(parts).backward()
        if red is not None:
            red.wait()

    want = []
    for i in range(gsteps):
        train_step(i, None)
        want.append([None if q.grad is None else q.grad.clone() for m in nets for q in m.parameters()])
    red = GradReducer([(f"net{k}", m) for k, m in enumerate(nets)], average=False)
    assert red.active and red.world == 1
    train_step(0, red)                           # communicator warm-up for the new message sizes
    torch.cuda.synchronize()
    got = []
    torch.cuda.set_sync_debug_mode("error")
    try:
        for i in range(gsteps):
            train_step(i, red)
            got.append([None if q.grad is None else q.grad.clone() for m in nets for q in m.parameters()])
            assert all(q.grad is None or q.grad.data_ptr() == red.view_of(q).data_ptr() for m in nets for q in m.parameters())
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    bad = sum(0 if ((a is None and b is None) or torch.equal(a, b)) else 1 for ws, gs in zip(want, got) for a, b in zip(ws, gs))
    out["grad_steps"] = gsteps
    out["grad_buckets"] = len(red.buckets)
    out["grad_collectives"] = red.issued
    out["grad_flat_bytes"] = int(red.flat.numel() * 4)
    out["grad_mismatching_tensors"] = bad
    assert bad == 0
    assert red.issued == (gsteps + 1) * len(red.buckets)
    assert float(sum(float(g.abs().sum()) for g in got[-1] if g is not None)) > 0
    dist.barrier()
    dist.destroy_process_group()
    out["ok"] = True
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
