#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MoCo-Flow volume-rendering hot path on MI355X.

Workload (BASELINE.json configs[1], "C2"): canonical NeRF (8x256, xyz F=10 -> 63, dir F=4 -> 27),
4096 rays x 64 samples per GPU, fp32 (exact-f32 MFMA), synthetic seeded rays and random-init
("dense" regime) weights, perturb = 0, noise_std = 0.  A "step" is one render_rays-equivalent
coarse pass over the batch = ONE launch of the fused kernel (mf_render_pass), inputs already
resident in HBM, outputs (rgb, depth, opacity) left on the device.

metric = ray-samples/s = (rays x samples evaluated by the network) / wall time, whole job.
  python bench.py                      # 1 GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W        # N ranks, weak scaling
With N > 1 every rank renders its own 4096-ray shard (rays are independent units) and the
per-batch loss partial sums [sum((rgb-gt)^2), count] are all-reduced over RCCL each step, asynchronously
(moco_flow_amd.dist.OverlappedLossReducer: the collective of step i overlaps step i+1's kernel).
(MF_BENCH_BACKEND=gloo MF_BENCH_SHARE_GPU=1: control-flow test of the N > 1 path on a single GPU.)

Extra JSON objects: "roofline" (MFMA bound: algorithmic Linear-layer FLOPs / measured kernel
time vs the 157.3 TFLOP/s fp32-matrix peak) and "cpu_baseline" (the CPU oracle, a PyTorch
restatement of the reference's op sequence, timed on this box's host cores, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_RAYS, N_SAMPLES = 4096, 64
FLOPS_PER_SAMPLE = {"nerf_dir": 1186816, "nerf_ind": 1181184, "nof_quat": 134400}
PEAK_F32_TFLOPS = 157.3          # MI355X_MICROARCH.md: fp32-input MFMA, 256 CU @ 2.4 GHz
PEAK_BF16_TFLOPS = 2516.0        # dense bf16 MFMA (only the hidden GEMMs run there in --precision bf16)


def build_models(dev, workload):
    import moco_flow_amd as M
    from moco_flow_amd import synth
    to_t = lambda sd: {k: torch.from_numpy(v) for k, v in sd.items()}
    if workload == "nerf":
        sd = synth.nerf_state(0, regime="dense")
        nerf = M.NeRF(8, 256, 63, [4], "dir", 27)
        nerf.load_state_dict(to_t(sd))
        return dict(embs=[M.Embedding(3, 10), None, M.Embedding(3, 4)], nerfs=[nerf.to(dev)], nof_embs=None,
                    nofs=None, states=dict(nerf=sd))
    sd = synth.nerf_state(0, extra_feat_type="ind", extra_feat_dim=5, regime="dense")
    nerf = M.NeRF(8, 256, 63, [4], "ind", 5)
    nerf.load_state_dict(to_t(sd))
    bw, fw = M.NoF(4, 128, 33, [2], "ind", 33, True), M.NoF(4, 128, 33, [2], "ind", 33, True)
    sb, sf = synth.nof_state(0, tag="bw", head_scale=0.25), synth.nof_state(0, tag="fw", head_scale=0.25)
    bw.load_state_dict(to_t(sb))
    fw.load_state_dict(to_t(sf))
    return dict(embs=[M.Embedding(3, 10), M.Embedding(1, 2), None], nerfs=[nerf.to(dev)],
                nof_embs=[M.Embedding(3, 5), M.Embedding(1, 16)], nofs=[bw.to(dev), fw.to(dev)],
                states=dict(nerf=sd, bw=sb, fw=sf))


def cpu_baseline(workload, states, rays_np, bg_np, budget_s=20.0):
    """The oracle (kind "port": PyTorch-CPU restatement of the reference op-for-op, pinned to the
    reference's golden vectors) on the same workload, host cores of this box."""
    from oracle import cpu_ref as R
    cores = torch.get_num_threads()
    rays, bg = torch.from_numpy(rays_np), torch.from_numpy(bg_np)
    if workload == "nerf":
        args = ([R.Embedding(3, 10), None, R.Embedding(3, 4)], [R.build_nerf(states["nerf"])])
        kw = dict(N_samples=N_SAMPLES, noise_std=0)
    else:
        args = ([R.Embedding(3, 10), R.Embedding(1, 2), None],
                [R.build_nerf(states["nerf"], extra_feat_type="ind", extra_feat_dim=5)])
        kw = dict(N_samples=N_SAMPLES, noise_std=0, nof_embeddings=[R.Embedding(3, 5), R.Embedding(1, 16)],
                  nof_models=[R.build_nof(states["bw"]), R.build_nof(states["fw"])], chain_local=True)
    with torch.no_grad():
        R.render_rays(rays[:256], bg[:256], *args, **kw)           # warm-up
        times = []
        t_start = time.perf_counter()
        while len(times) < 5 and (time.perf_counter() - t_start) < budget_s:
            t0 = time.perf_counter()
            out = R.render_rays(rays, bg, *args, **kw)
            times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    cpu = "unknown CPU"
    try:
        with open("/proc/cpuinfo") as fh:
            cpu = next(l.split(":", 1)[1].strip() for l in fh if l.startswith("model name"))
    except (OSError, StopIteration):
        pass
    return dict(value=N_RAYS * N_SAMPLES / med, unit="ray-samples/s", cores=cores, kind="port",
                sample=f"{len(times)} full {N_RAYS}x{N_SAMPLES} batches, median {med:.3f} s/batch, "
                       f"torch {torch.__version__} CPU fp32, {cores} threads on {cpu}"), out


def train_leg(M, models, rays, bg, gt, kw, workload, steps=10):
    """fwd + bwd of the same batch (reported beside, never instead of, the forward metric):
    render_rays -> MSE -> backward through the HIP kernels (training forward with dump, mf_nerf_backward,
    mf_weight_grads; NoF evaluations through mf_nof_points_dump / mf_nof_backward)."""
    mods = list(models["nerfs"]) + list(models["nofs"] or [])
    crit = M.get_loss(dict(type="MSE"))

    def one():
        for m in mods:
            m.zero_grad(set_to_none=True)
        res = M.render_rays(rays, bg, models["embs"], models["nerfs"], **kw)
        loss = crit(res, gt)
        if "nof_local_disp_coarse" in res:
            loss = loss + 0.1 * res["nof_local_disp_coarse"].mean()
        loss.backward()

    one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    # Linear-layer MACs x 2: forward + weight gradients (593 408 each) + input-gradient chain (557 696)
    flops = 2 * (593408 * 2 + 557696) if workload == "nerf" else None
    out = {"value": N_RAYS * N_SAMPLES / (ms * 1e-3), "unit": "ray-samples/s", "ms_per_step": ms, "steps": steps,
           "what": "render_rays + MSELoss + backward (all parameter gradients), same batch as the headline"}
    if flops:
        out["flops_per_sample"] = flops
        out["mfma_frac"] = N_RAYS * N_SAMPLES * flops / (ms * 1e-3) / 1e12 / PEAK_F32_TFLOPS
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=["nerf", "moco"], default="nerf",
                    help="nerf = BASELINE config C2 (headline); moco = C3-shaped chain in fp32")
    ap.add_argument("--precision", choices=["f32", "bf16"], default="f32",
                    help="f32 = exact-fp32 MFMA (headline, config C2); bf16 = bf16 hidden GEMMs (configs C3-C5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-leg", action="store_true",
                    help="skip the extra fwd+bwd measurement (SURVEY.md §8d: reported separately from the graded forward)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit(f"--gpus {a.gpus} needs `python -m torch.distributed.run --nproc-per-node {a.gpus} bench.py ...`")
        a.gpus = world
    if os.environ.get("MF_BENCH_SHARE_GPU"):                          # control-flow tests: every rank on GPU 0
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("MF_BENCH_BACKEND", "nccl")           # "nccl" is RCCL on ROCm; "gloo": control-flow tests only
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    import moco_flow_amd as M
    from moco_flow_amd import rendering, synth
    rendering.STRICT_RNG = False        # noise_std = 0: do not launch the reference's dead randn
    rendering.set_precision(a.precision)
    M._lib.lib()                        # fail loudly if the HIP library is missing

    models = build_models(dev, a.workload)
    # weak scaling: each rank owns a contiguous block of the global batch (global ray order kept)
    rays_np, bg_np = synth.rays(0, N_RAYS * world, chained=False)
    lo = rank * N_RAYS
    rays = torch.from_numpy(rays_np[lo:lo + N_RAYS]).to(dev)
    bg = torch.from_numpy(bg_np[lo:lo + N_RAYS]).to(dev)
    gt = torch.from_numpy(synth.uniform01(123 + rank, N_RAYS * 3).reshape(N_RAYS, 3).astype(np.float32)).to(dev)
    kw = dict(N_samples=N_SAMPLES, noise_std=0, perturb=0)
    if a.workload == "moco":
        kw.update(nof_embeddings=models["nof_embs"], nof_models=models["nofs"], chain_local=True)

    # per-step loss partials [sum (rgb - gt)^2, count], all-reduced over RCCL without serialising with the
    # next step's launch (moco_flow_amd.dist.OverlappedLossReducer)
    from moco_flow_amd.dist import OverlappedLossReducer
    reducer = OverlappedLossReducer(2, dev) if dist is not None else None
    part = torch.zeros(2, device=dev, dtype=torch.float64)

    def step():
        out = M.render_rays(rays, bg, models["embs"], models["nerfs"], **kw)
        if reducer is not None:
            d = out["rgb_coarse"] - gt
            part[0] = (d * d).sum()
            part[1] = d.numel()
            reducer.push(part)                                        # RCCL over xGMI, 16 bytes, asynchronous
        return out

    with torch.no_grad():
        for _ in range(a.warmup):
            out = step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            ev[i][0].record()
            out = step()
            ev[i][1].record()
        if reducer is not None:
            reducer.finish()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = float(np.mean([s.elapsed_time(e) for s, e in ev]))   # per-launch span on the launch stream

    samples = N_RAYS * N_SAMPLES * world * a.steps
    value = samples / elapsed
    if a.workload == "nerf":
        flops_per_sample = FLOPS_PER_SAMPLE["nerf_dir"]
    else:
        flops_per_sample = FLOPS_PER_SAMPLE["nerf_ind"] + 2 * FLOPS_PER_SAMPLE["nof_quat"]
    achieved = N_RAYS * N_SAMPLES * flops_per_sample / (kernel_ms * 1e-3) / 1e12
    peak = PEAK_F32_TFLOPS if a.precision == "f32" else PEAK_BF16_TFLOPS
    line = {
        "metric": "ray-samples/sec (4096 rays x 64 samples)", "value": value, "unit": "ray-samples/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
        "config": {"workload": ("C2: canonical NeRF 8x256 (xyz F=10, dir F=4), fused HIP encode+MLP+composite, "
                                "fp32 MFMA" if a.workload == "nerf" else
                                "C3-shaped: bw NoF -> NeRF(ind) -> fw NoF local chain") + f" [{a.precision}]",
                   "rays_per_gpu": N_RAYS, "samples_per_ray": N_SAMPLES, "global_rays": N_RAYS * world,
                   "sharding": f"rays{world}" if world > 1 else "none"},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                     "frac": achieved / peak,
                     # HBM bytes per launch from the committed PMC passes of this same command
                     # (profiles/r01f_summary.txt: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE); C2 only
                     "traffic": 26.2e6 if (a.workload == "nerf" and a.precision == "f32") else None,
                     "traffic_unit": "B/launch",
                     "kernel_ms": kernel_ms, "flops_per_launch": N_RAYS * N_SAMPLES * flops_per_sample},
    }
    if rank == 0 and world == 1 and not a.no_train_leg and a.precision == "f32":
        line["fwd_bwd"] = train_leg(M, models, rays, bg, gt, kw, a.workload)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        base, ref = cpu_baseline(a.workload, models["states"], rays_np[:N_RAYS], bg_np[:N_RAYS])
        line["cpu_baseline"] = base
        errs, l2 = {}, {}
        for k, v in ref.items():
            if k.startswith("nof_"):
                continue
            g = out[k].cpu().double()
            errs[k] = float((g - v.double()).abs().max() / v.double().abs().max())
            l2[k] = float((g - v.double()).norm() / v.double().norm())
        mse = float(((out["rgb_coarse"].cpu().double() - ref["rgb_coarse"].double()) ** 2).mean())
        line["error_vs_cpu"] = {"max_rel": errs, "l2_rel": l2, "psnr_equiv_db": (-10 * np.log10(mse)) if mse > 0 else float("inf")}
        line["speedup_vs_cpu"] = value / base["value"]
    if rank == 0:
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
