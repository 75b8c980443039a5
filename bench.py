#!/usr/bin/env python3
"""bench.py -- benchmark of the MoCo-Flow volume-rendering hot path on MI355X.

    python bench.py                                   # 1 GPU, headline (BASELINE config C2)
    python bench.py --gpus N [--steps K --warmup W]   # N ranks on one node; run BARE it spawns its own
                                                      # N workers (one per GPU, RCCL), or run it under
                                                      # `python -m torch.distributed.run --nproc-per-node N`
    python bench.py --config C3|C3g|C4|C5             # make another BASELINE config the main line

A "step" is one render_rays-equivalent call over the rank's batch, inputs resident in HBM, outputs left on the
device.  metric = ray-samples/s = network-evaluated samples of all ranks / wall time (max over ranks).

BASELINE.json configs:
  C2   canonical NeRF 8x256 (xyz F=10, dir F=4), fp32 (exact-f32 MFMA), 4096 rays x 64 samples -- THE HEADLINE:
       the main line at every N (4096 rays per rank, weak scaling, NO loss path / collective: that is C4's), so that
       the driver's 1/2/4/8-GPU curve compares like with like.
  C3   bw NoF -> NeRF(ind) -> fw NoF local chain, bf16 hidden GEMMs, 4096 x 64      (C3g: + global chain)
  C4   C3 ray-sharded, 4096 rays per rank, per-batch loss partials all-reduced over RCCL (N > 1)
  C5   coarse 64 + fine 128 with the inverse-CDF resample, MoCo local+global chains, bf16, 1024 rays per rank
       (8192 rays on 8 GPUs)
The default run also measures the other configs as short extra legs and reports them in the same JSON line under
"configs" (N = 1: C3, C3g, C5 shard, and C2x / C3x / C5x = the same workloads in bf16x3, the fp32-class mode of the bf16
matrix pipe -- every product as three bf16 products of (hi, lo) operand pairs; N > 1: C4, C5), each with its own
roofline fraction and error against the CPU oracle, so every BASELINE config is driver-measured without changing what
`value` means.  The legs are sub-millisecond passes: each gets 100 warm-up and 200 timed steps of its own (a handful of
warm-up steps leaves the clocks ramping); the main line uses exactly the --steps / --warmup it was given.  The timed region
of every config holds the calls alone: the per-call event pairs behind "step_span_ms" are recorded in a second, untimed loop,
and the interpreter's cyclic garbage collector is collected + frozen in front of it and off inside it (as `timeit` does): one
full collection of the heap torch's import leaves behind stalls the host 40-50 ms at a call of its own choosing, which a
20-step region of a 0.34 ms pass reports as 2.3 ms per step (MF_BENCH_STEP_TIMES=2 prints each call's host time).

Extra JSON objects: "roofline" (MFMA bound: algorithmic Linear-layer FLOPs of the dominant kernel's launch / its
average duration, HIP events on the launch stream around the launch alone, vs the dense matrix peak of the dtype;
"step_span_ms" is the whole render_rays call incl. resample / consensus compaction) and "cpu_baseline" (the CPU oracle -- a PyTorch restatement of the
reference's op sequence, pinned to the reference's golden vectors -- timed on this box's host cores with the
best thread count of a sweep; rank 0, N = 1 only).  The line ENDS with "legs": one compact record per measured config
(ms per step, kernel ms, roofline fraction, PSNR-equivalent dB, worst max-rel) + the training-step times.
"""
import argparse
import gc
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOPS = {"nerf_dir": 1186816, "nerf_ind": 1181184, "nof_quat": 134400}   # per sample, SURVEY.md §8d
# TFLOP/s, MI355X_MICROARCH.md: fp32-input MFMA / dense bf16 MFMA.  bf16x3 issues SEVERAL bf16 matrix instructions per
# algorithmic product -- three for the NeRF (hi*hi + hi*lo + lo*hi of bf16 (hi, lo) operand pairs) and, since round 5, three for
# the NoF (IEEE-half (hi, lo) pairs on the f16 instruction of the same rate; rounds 3-4: six, bf16 triples) -- so its ceiling in
# ALGORITHMIC FLOP/s is the bf16 peak / 3: peak_of().
PEAK = {"f32": 157.3, "bf16": 2516.0}
HBM_PEAK_TBS = 8.0          # MI355X_MICROARCH.md: HBM3E, the roofline of the training step's streaming kernels (legs.aux.wgrad)
X3_PRODUCTS = {"nerf": 3, "nof": 3}
PEAK_NOTE = {"f32": "dense fp32-input MFMA peak", "bf16": "dense bf16 MFMA peak",
             "bf16x3": "dense bf16 MFMA peak x algorithmic / issued FLOP = peak / 3 (three matrix instructions per product: bf16 (hi, lo) "
                       "pairs in the NeRF, IEEE-half (hi, lo) pairs in the NoF; `achieved` counts algorithmic FLOP)"}


def peak_of(cfg):
    if cfg["precision"] != "bf16x3":
        return PEAK[cfg["precision"]]
    nerf, nof = FLOPS["nerf_" + cfg["net"]], n_nof(cfg) * FLOPS["nof_quat"]
    return PEAK["bf16"] * (nerf + nof) / (X3_PRODUCTS["nerf"] * nerf + X3_PRODUCTS["nof"] * nof)


CONFIGS = {
    "C2": dict(short="C2 NeRF 8x256 fp32 4096x64", net="dir", precision="f32", rays=4096, S=64, M=0, nof=None,
               what="C2: canonical NeRF 8x256 (xyz F=10, dir F=4), fused HIP encode+MLP+composite, fp32 MFMA"),
    # the main line at N > 1 (north_star: "partition batches across the 8 GPUs ... with an RCCL all-reduce of the pixel/loss
    # tensor"): the SAME kernel, dtype and per-rank batch as C2, + the MSE partials of the rank's pixels (mf_loss_partials) and
    # their 96-byte all-reduce inside the step -- so the driver's 1 -> N curve compares like with like AND carries the collective
    "C2dp": dict(short="C2 NeRF 8x256 fp32 4096x64/rank + loss all-reduce", net="dir", precision="f32", rays=4096, S=64, M=0, nof=None, loss=True, profile="C2",
                 what="C2 ray-sharded: canonical NeRF 8x256 fp32, 4096 rays x 64 per rank + RCCL all-reduce of the per-batch loss partials"),
    "C2b": dict(short="C2 shape bf16", net="dir", precision="bf16", rays=4096, S=64, M=0, nof=None,
                what="C2 shape in bf16 (canonical NeRF, bf16 hidden GEMMs; a kernel-tuning leg, not a BASELINE config)"),
    "C3f": dict(short="C3 chain fp32", net="ind", precision="f32", rays=4096, S=64, M=0, nof="local",
                what="C3 chain in fp32 (bw NoF -> NeRF(ind) -> fw NoF, exact-fp32 MFMA; a kernel-tuning leg, not a BASELINE config)"),
    "C3": dict(short="C3 MoCo local bf16 4096x64", net="ind", precision="bf16", rays=4096, S=64, M=0, nof="local",
               what="C3: bw NoF -> NeRF(ind) -> fw NoF local consensus chain, bf16 hidden GEMMs"),
    "C2x": dict(short="C2 bf16x3", net="dir", precision="bf16x3", rays=4096, S=64, M=0, nof=None,
                what="C2 in the contract mode of the bf16 pipe (bf16x3: every matrix product as three bf16 products of (hi, lo) "
                     "operand pairs, fp32 accumulation and heads)"),
    "C3x": dict(short="C3 bf16x3", net="ind", precision="bf16x3", rays=4096, S=64, M=0, nof="local",
                what="C3 in the contract mode of the bf16 pipe (bf16x3: the NeRF's products as three bf16 products of (hi, lo) pairs, "
                     "the NoFs' as three of IEEE-half (hi, lo) pairs, fp32 accumulation, heads and per-ray image-index bias: 1e-4 max-rel)"),
    "C5x": dict(short="C5 shard bf16x3", net="ind", precision="bf16x3", rays=1024, S=64, M=128, nof="global", loss=True,
                what="C5 in the contract mode of the bf16 pipe (bf16x3)"),
    "C3g": dict(short="C3 + global chain bf16", net="ind", precision="bf16", rays=4096, S=64, M=0, nof="global",
                what="C3 + global chain (5 NoF evaluations per sample), bf16 hidden GEMMs"),
    "C4": dict(short="C4 = C3/rank + loss all-reduce", net="ind", precision="bf16", rays=4096, S=64, M=0, nof="local", loss=True,
               what="C4: C3 ray-sharded (4096 rays per rank) + all-reduce of the per-batch loss partials"),
    "C5": dict(short="C5 shard 1024x(64+192) bf16", net="ind", precision="bf16", rays=1024, S=64, M=128, nof="global", loss=True,
               what="C5: coarse 64 + fine 128 (inverse-CDF resample), MoCo local+global chains, bf16, 1024 rays per rank"),
    # BASELINE config 5 at its FULL size on one GPU (the largest single-GPU configuration of `configs`): 8192 rays, not the
    # 1024-ray per-rank shard -- 32 coarse / 96 fine 256-sample tiles per CU instead of one / three
    "C5full": dict(short="C5 8192x(64+192) bf16", net="ind", precision="bf16", rays=8192, S=64, M=128, nof="global", loss=True, steps=(40, 10),
                   what="C5 at full size on ONE GPU: 8192 rays, coarse 64 + fine 128 (inverse-CDF resample), MoCo local+global chains, bf16"),
    "C5xfull": dict(short="C5 8192x(64+192) bf16x3", net="ind", precision="bf16x3", rays=8192, S=64, M=128, nof="global", loss=True, steps=(20, 5),
                    what="C5 at full size on ONE GPU in the contract mode of the bf16 pipe (bf16x3)"),
}


def n_nof(cfg):
    return {None: 0, "bw": 1, "local": 2, "global": 5}[cfg["nof"]]


def flops_per_sample(cfg):
    return FLOPS["nerf_" + cfg["net"]] + n_nof(cfg) * FLOPS["nof_quat"]


def samples_per_ray(cfg):
    return cfg["S"] + ((cfg["S"] + cfg["M"]) if cfg["M"] else 0)


# ------------------------------------------------------------------ self-spawn (parent: no GPU call, ever)
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_workers(n, argv):
    """`bench.py --gpus N` invoked bare: start N fresh worker processes (this same file, RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment) BEFORE anything here touches the GPU; rank 0 prints the JSON
    line on the inherited stdout.  Non-zero exit if any worker fails (the others are then terminated)."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), MF_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    alive = list(procs)
    while alive:
        time.sleep(0.05)
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in alive:          # exact PIDs we started
                    q.terminate()
    return rc


# ------------------------------------------------------------------ worker
def build_models(M, synth, dev, cfg):
    import torch
    to_t = lambda sd: {k: torch.from_numpy(v) for k, v in sd.items()}
    dims = {"dir": 27, "ind": 5}[cfg["net"]]
    states, nerfs = {}, []
    for tag in (["coarse", "fine"] if cfg["M"] else ["coarse"]):
        sd = synth.nerf_state(0, extra_feat_type=cfg["net"], extra_feat_dim=dims, regime="dense",
                              tag="nerf" if tag == "coarse" else "nerf_fine")
        m = M.NeRF(8, 256, 63, [4], cfg["net"], dims)
        m.load_state_dict(to_t(sd))
        nerfs.append(m.to(dev) if dev is not None else m)
        states[tag] = sd
    embs = [M.Embedding(3, 10), M.Embedding(1, 2) if cfg["net"] == "ind" else None,
            M.Embedding(3, 4) if cfg["net"] == "dir" else None]
    nof_embs = nofs = None
    if cfg["nof"]:
        nof_embs, nofs = [M.Embedding(3, 5), M.Embedding(1, 16)], []
        for tag in ("bw", "fw"):
            sd = synth.nof_state(0, tag=tag, head_scale=0.25)
            m = M.NoF(4, 128, 33, [2], "ind", 33, True)
            m.load_state_dict(to_t(sd))
            nofs.append(m.to(dev) if dev is not None else m)
            states[tag] = sd
    return dict(embs=embs, nerfs=nerfs, nof_embs=nof_embs, nofs=nofs, states=states)


def render_kwargs(cfg, models):
    kw = dict(N_samples=cfg["S"], N_importance=cfg["M"], noise_std=0, perturb=0)
    if cfg["nof"]:
        kw.update(nof_embeddings=models["nof_embs"], nof_models=models["nofs"],
                  chain_local=cfg["nof"] in ("local", "global"), chain_global=cfg["nof"] == "global")
    return kw


def oracle_render(cfg, states, rays, bg, z_fine=None):
    """The oracle on the same workload (checker / cpu_baseline leg only).  ``z_fine``: evaluate the fine pass on
    the given (the HIP path's own) fine depths, so that the comparison is sample-for-sample."""
    import torch
    from oracle import cpu_ref as R
    dims = {"dir": 27, "ind": 5}[cfg["net"]]
    nerfs = [R.build_nerf(states["coarse"], extra_feat_type=cfg["net"], extra_feat_dim=dims)]
    if cfg["M"]:
        nerfs.append(R.build_nerf(states["fine"], extra_feat_type=cfg["net"], extra_feat_dim=dims))
    embs = [R.Embedding(3, 10), R.Embedding(1, 2) if cfg["net"] == "ind" else None,
            R.Embedding(3, 4) if cfg["net"] == "dir" else None]
    kw = dict(N_samples=cfg["S"], N_importance=cfg["M"], noise_std=0, perturb=0)
    if cfg["nof"]:
        kw.update(nof_embeddings=[R.Embedding(3, 5), R.Embedding(1, 16)],
                  nof_models=[R.build_nof(states["bw"]), R.build_nof(states["fw"])],
                  chain_local=cfg["nof"] in ("local", "global"), chain_global=cfg["nof"] == "global")
    if z_fine is not None:
        kw["_z_fine_override"] = z_fine
    with torch.no_grad():
        return R.render_rays(rays, bg, embs, nerfs, **kw)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            return next(l.split(":", 1)[1].strip() for l in fh if l.startswith("model name"))
    except (OSError, StopIteration):
        return "unknown CPU"


def cpu_baseline(cfg, states, rays_np, bg_np, budget_s=30.0):
    """kind "port": the oracle timed on this box's host cores.  Thread sweep first (one batch each), then the
    median of up to 3 more batches at the fastest thread count; bounded by `budget_s` of CPU work."""
    import numpy as np
    import torch
    rays, bg = torch.from_numpy(rays_np), torch.from_numpy(bg_np)
    n = rays.shape[0]
    ncpu = os.cpu_count() or 8
    sweep = sorted({t for t in (16, 32, 64) if t <= ncpu} or {min(ncpu, 64)})      # (round 5: 8 / 128 never won on the 64-core box)
    t_start = time.perf_counter()
    oracle_render(cfg, states, rays[:128], bg[:128])                  # warm-up
    per_thread, out = {}, None
    for t in sweep:
        if per_thread and time.perf_counter() - t_start > budget_s * 0.6:
            break
        torch.set_num_threads(t)
        t0 = time.perf_counter()
        out = oracle_render(cfg, states, rays, bg)
        per_thread[t] = time.perf_counter() - t0
    best = min(per_thread, key=per_thread.get)
    torch.set_num_threads(best)
    times = [per_thread[best]]
    while len(times) < 4 and time.perf_counter() - t_start < budget_s:
        t0 = time.perf_counter()
        out = oracle_render(cfg, states, rays, bg)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    sweep_s = ", ".join(f"{t}t {s:.2f}s" for t, s in per_thread.items())
    return dict(value=n * samples_per_ray(cfg) / med, unit="ray-samples/s", cores=best, kind="port", cpu=cpu_model(),
                sample_short=f"{len(times)} full batches, {'/'.join(map(str, per_thread))}t sweep",
                sample=f"{len(times)} full {n}x{samples_per_ray(cfg)} batches at the fastest of a thread sweep "
                       f"({sweep_s}), median {med:.3f} s/batch, torch {torch.__version__} CPU fp32, "
                       f"{best} threads on {cpu_model()} ({ncpu} logical CPUs)"), out


def errors_vs(ref, out):
    import numpy as np
    errs, l2 = {}, {}
    for k, v in ref.items():
        if k.startswith("nof_") or k not in out:
            continue
        g, v = out[k].detach().cpu().double(), v.double()
        errs[k] = float((g - v).abs().max() / v.abs().max())
        l2[k] = float((g - v).norm() / v.norm())
    key = "rgb_fine" if "rgb_fine" in ref else "rgb_coarse"
    mse = float(((out[key].detach().cpu().double() - ref[key].double()) ** 2).mean())
    return {"max_rel": errs, "l2_rel": l2, "psnr_equiv_db": (-10 * np.log10(mse)) if mse > 0 else float("inf"),
            "psnr_key": key}


def traffic_of(name):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes of this same command
    (profiles/traffic.json, written by tools/summarize_prof.py: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE);
    null when no profile of this configuration is committed.  Never a number invented here."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            t = json.load(fh).get(name)
        if t:
            return t.get("bytes_per_launch"), t.get("source"), t
    except (OSError, ValueError, KeyError):
        pass
    return None, None, {}


def train_traffic(key):
    t = traffic_of(key)[2]
    return {"hbm_gb_per_step": t["hbm_gb_per_step"], "hbm_profile": t.get("profile")} if t.get("hbm_gb_per_step") else {}


def algorithmic_bytes(cfg, launch_samples):
    """HBM bytes ONE launch of the dominant kernel must move (48 B in + 20 B out per ray; + the weights once;
    + 4 B per sample and (N, S) plane the caller asked for: a consensus pass hands back alphas and one distance plane per
    chain; + the bf16 modes' per-ray NoF bias rows, 512 B per (ray, network-index combination, embedded layer), written by
    mf_render_prepare and read once).  Weights in the element size of the packed stream (f32 4 B, bf16 2 B, bf16x3 2 x 2 B
    for the NeRF / 3 x 2 B for the NoFs)."""
    S = cfg["S"] + cfg["M"] if cfg["M"] else cfg["S"]
    rays = launch_samples // S
    n_nerf = {"dir": 595844, "ind": 593028}[cfg["net"]]
    wbytes = {"f32": (4, 4), "bf16": (2, 2), "bf16x3": (4, 6)}[cfg["precision"]]
    w = n_nerf * wbytes[0] + (2 if cfg["nof"] in ("local", "global") else (1 if cfg["nof"] else 0)) * 67721 * wbytes[1]
    planes = {None: 0, "bw": 0, "local": 2, "global": 3}[cfg["nof"]]
    bias = 0
    if cfg["nof"] and cfg["precision"] != "f32":
        bias = rays * {"bw": 1, "local": 2, "global": 4}[cfg["nof"]] * 2 * 512
    return rays * 68 + w + planes * 4 * launch_samples + bias


def train_leg(M, torch, models, rays, bg, gt, kw, cfg, steps=10):
    """fwd + bwd of the same batch (reported beside, never instead of, the forward metric)."""
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
    n = rays.shape[0] * samples_per_ray(cfg)
    out = {"value": n / (ms * 1e-3), "unit": "ray-samples/s", "ms_per_step": ms, "steps": steps,
           "what": "render_rays + MSELoss + backward (all parameter gradients), same batch as the main line"}
    if cfg["nof"] is None and cfg["net"] == "dir":
        # Linear-layer MACs x 2: forward + weight gradients (593 408 each) + input-gradient chain (557 696)
        flops = 2 * (593408 * 2 + 557696)
        from moco_flow_amd import autograd as A
        out["flops_per_sample"] = flops
        out["wgrad_precision"] = A.WGRAD_PRECISION
        out["dx_precision"] = A.DX_PRECISION
        # (no single roofline fraction here: the forward runs on the fp32 matrix pipe, the dX chain and the large weight-gradient
        #  blocks on the bf16 pipe (three products) and against HBM -- a sum over one pipe's peak is not a utilisation)
    return out


def train_shape_legs(M, synth, torch, dev, steps=6):
    """The reference's two training-step shapes (reported beside the forward metric), forward +
    backward through the drop-in exactly as the unchanged trainer calls it, median of `steps` individually timed steps:
      stage1: init_nerf.yaml -- N_rand 5120 rays x (128 coarse + 256 fine) samples, NeRF(dir / 27) x 2, MSE loss
              (trainer_nerf.py:149-169);
      joint:  c2f.yaml -- 1024 rays x (128 + 256), NeRF(ind / 5) x 2 behind the backward NoF, local + global consensus
              chains, perturb = 1, MSE + consensus means (trainer_moco_flow.py:200-216, 317-328)."""
    from moco_flow_amd import autograd as A, rendering

    def load(m, sd):
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        return m.to(dev)

    def med(f):
        import gc
        f()
        f()                                  # (two warm-up steps: the first one after empty_cache() goes to the driver for every dump)
        torch.cuda.synchronize()
        gc.collect()
        gc.disable()                         # (a collection inside a 15 ms step is a visible fraction of it)
        try:
            ts = []
            for _ in range(steps + 2):
                t0 = time.perf_counter()
                f()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) * 1e3)
        finally:
            gc.enable()
        return float(sorted(ts)[len(ts) // 2])

    crit = M.get_loss(dict(type="MSE"))
    out = {"dx_precision": A.DX_PRECISION, "wgrad_precision": A.WGRAD_PRECISION, "train_forward_precision": rendering.TRAIN_FORWARD_PRECISION,
           "what": "render_rays + loss + backward (every parameter gradient), median ms per step; synthetic rays, dense random weights"}
    # stage 1
    N = 5120
    nerfs = [load(M.NeRF(8, 256, 63, [4], "dir", 27), synth.nerf_state(0, regime="dense", tag=t)) for t in ("coarse", "fine")]
    embs = [M.Embedding(3, 10), None, M.Embedding(3, 4)]
    r, b = synth.rays(0, N)
    rays, bg = torch.from_numpy(r).to(dev), torch.from_numpy(b).to(dev)
    gt = torch.rand(N, 3, device=dev)
    kw = dict(N_samples=128, N_importance=128, noise_std=0, perturb=0)

    def stage1():
        for m in nerfs:
            m.zero_grad(set_to_none=True)
        crit(M.render_rays(rays, bg, embs, nerfs, **kw), gt).backward()

    # hbm_gb_per_step: FETCH_SIZE x 2 + WRITE_SIZE summed over the step's kernels, from the committed rocprofv3 passes of the same step
    # (profiles/traffic.json["train_stage1" | "train_joint"], tools/profile_train_all.sh): a profile-derived constant like roofline.from_profiles
    out["stage1"] = {"ms_per_step": med(stage1), "rays": N, "samples_per_ray": 384, **train_traffic("train_stage1")}
    # the same step with the opt-in three-product training forward (set_train_forward_precision("bf16x3"): forward values to
    # 5e-6 max-rel, but gradients then differ from the fp32 oracle's by ~3e-3 max-rel through ReLU units that change side
    # -- outside the 1e-4 bars the default is held to; reported for information)
    prev = rendering.TRAIN_FORWARD_PRECISION
    rendering.set_train_forward_precision("bf16x3")
    try:
        out["stage1_optin_bf16x3_forward"] = {"ms_per_step": med(stage1), "rays": N, "samples_per_ray": 384,
                                             "note": "not the default: gradient parity 3e-3 instead of 1e-4 (ReLU mask flips)"}
    finally:
        rendering.set_train_forward_precision(prev)
    del nerfs, rays, bg, gt
    torch.cuda.empty_cache()
    # joint MoCo stage
    N = 1024
    nerfs, nofs, rays, bg, gt, embs, kw = joint_stage_setup(M, synth, torch, dev, N)

    def joint():
        for m in nerfs + nofs:
            m.zero_grad(set_to_none=True)
        res = M.render_rays(rays, bg, embs, nerfs, **kw)
        loss = crit(res, gt)
        for k in ("nof_local_disp_coarse", "nof_global_disp_coarse", "nof_local_disp_fine", "nof_global_disp_fine"):
            loss = loss + 0.1 * res[k].mean()
        loss.backward()

    out["joint"] = {"ms_per_step": med(joint), "rays": N, "samples_per_ray": 384, **train_traffic("train_joint")}
    # the whole training ITERATION as a trainer runs it: + the optimizer step (Adam over the four networks), after which every
    # network re-packs its weight streams at the next forward (round 5: +0.26 ms)
    opt = torch.optim.Adam([q for m in nerfs + nofs for q in m.parameters()], lr=1e-7)

    def joint_iteration():
        joint()
        opt.step()

    out["joint_with_optimizer"] = {"ms_per_step": med(joint_iteration), "rays": N, "samples_per_ray": 384, "optimizer": "Adam"}
    del opt
    # round 5: the same step with the opt-in three-product training forward, now for passes WITH NoF too
    # (render_kernel_bf16<true, true, true>; forward values to 1e-4, gradients inside the fp32 oracle's own noise floor against
    # the float64 truth, tests/test_gpu_parity.py::test_train_forward_bf16x3_moco) -- reported beside the default, never instead
    torch.cuda.empty_cache()          # (a step holds ~20 GB of dumps: differently shaped blocks on top of the cached ones made the allocator go to the driver every step)
    rendering.set_train_forward_precision("bf16x3")
    try:
        out["joint_optin_bf16x3_forward"] = {"ms_per_step": med(joint), "rays": N, "samples_per_ray": 384,
                                            "note": "not the default: ReLU mask flips put gradient parity at the noise-floor yardstick instead of 1e-4"}
    finally:
        rendering.set_train_forward_precision(prev)
    del nerfs, nofs, rays, bg, gt
    torch.cuda.empty_cache()
    return out


def joint_stage_setup(M, synth, torch, dev, N=1024, seed=0):
    """The joint MoCo stage's step shape (c2f.yaml: N rays x (128 + 128 + 128), NeRF(ind / 5) x 2 behind the backward NoF,
    local + global consensus chains, perturb = 1) -> (networks, rays, bg, gt, embeddings, render kwargs)."""
    def load(m, sd):
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        return m.to(dev)
    nerfs = [load(M.NeRF(8, 256, 63, [4], "ind", 5), synth.nerf_state(0, extra_feat_type="ind", extra_feat_dim=5, regime="dense", tag=t))
             for t in ("coarse", "fine")]
    nofs = [load(M.NoF(4, 128, 33, [2], "ind", 33, True), synth.nof_state(0, use_quat=True, tag=t, head_scale=0.25)) for t in ("bw", "fw")]
    embs = [M.Embedding(3, 10), M.Embedding(1, 2), None]
    r, b = synth.rays(seed, N, chained=True)
    rays, bg = torch.from_numpy(r).to(dev), torch.from_numpy(b).to(dev)
    gt = torch.rand(N, 3, device=dev)
    kw = dict(nof_embeddings=[M.Embedding(3, 5), M.Embedding(1, 16)], nof_models=nofs, chain_local=True, chain_global=True,
              N_samples=128, N_importance=128, noise_std=0, perturb=1.0)
    return nerfs, nofs, rays, bg, gt, embs, kw


def train_dp_leg(M, synth, torch, dev, dist, rank, world, steps=6):
    """Data-parallel TRAINING across the ranks (training half): the joint stage's step, 1024 rays per rank
    (weak scaling), global loss from the all-reduced 12 partials (dist.global_partials: 96 B), the four networks' flat
    gradient (5.3 MB) all-reduced per network from the post-accumulate hooks while the remaining backward launches run
    (dist.GradReducer) -- median ms per step with and without the reduce (MAX over ranks)."""
    from moco_flow_amd import dist as D, losses
    nerfs, nofs, rays, bg, gt, embs, kw = joint_stage_setup(M, synth, torch, dev, 1024, seed=rank)
    nets = nerfs + nofs

    def total_of(parts):
        t = losses.from_partials(parts)
        return t["img_loss"] + 0.1 * t["nof_local"] + 0.1 * t["nof_global"]

    def step(red):
        for m in nets:
            m.zero_grad(set_to_none=True)
        res = M.render_rays(rays, bg, embs, nerfs, _loss_target=gt, **kw)
        total_of(res["loss_partials"] if red is None else D.global_partials(res["loss_partials"])).backward()
        if red is not None:
            red.wait()

    def med(red):
        step(red)
        step(red)
        torch.cuda.synchronize()
        ts = []
        for _ in range(steps):
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            step(red)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        t = torch.tensor([sorted(ts)[len(ts) // 2]], device=dev, dtype=torch.float64)
        if dist is not None:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    local = med(None)
    red = D.GradReducer([(n, m) for n, m in zip(("nerf_coarse", "nerf_fine", "nof_bw", "nof_fw"), nets)], average=False)
    with_reduce = med(red)
    red.remove()
    return {"ms_per_step_local": local, "ms_per_step_allreduce": with_reduce, "rays_per_gpu": 1024, "samples_per_ray": 384,
            "flat_gradient_bytes": int(red.flat.numel() * 4), "buckets": len(red.buckets), "world": world,
            "what": "joint MoCo training step per rank: without any collective / with the global loss from the all-reduced "
                    "partials + the per-network gradient all-reduce overlapped with backward (dist.GradReducer), median ms, MAX over ranks"}


def aux_legs(M, synth, torch, dev):
    """The paths either side of render_rays that SURVEY.md 8(f) rows 2-3 name, driver-visible:
      lattice: the mesh-extraction sigma query (test.py:55-56,96 -> visualize_mesh, trainer_moco_flow.py:485-548) on a 256^3
               lattice as ONE fused launch (mf_points_sigma_p): canonical space in f32 / bf16x3, observation space (bw NoF
               first) in bf16x3; points/s and the fraction of the matrix peak (NeRF sigma path 982 528 FLOP per point,
               + 134 400 through the NoF; bf16x3: peak / 3 resp. the NeRF-3 / NoF-6 mix);
      image:   one 512 x 512 image.render_image call (MoCoFlowTrainer.render, trainer_moco_flow.py:226-268): rays made on the
               device, 1/7 of the pixels masked out, MoCo path (bw NoF -> NeRF), 64 + 128 samples, test_time, bf16."""
    import functools
    import numpy as np
    from moco_flow_amd import camera, image, rendering

    def load(m, sd):
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        return m.to(dev)

    def timeit(f, n=3):
        f()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n

    sig = lambda x, n=3: float(f"{x:.{n}g}")
    out = {}
    with torch.no_grad():
        nerf = load(M.NeRF(8, 256, 63, [4], "ind", 5), synth.nerf_state(0, extra_feat_type="ind", extra_feat_dim=5, regime="dense"))
        nof = load(M.NoF(4, 128, 33, [2], "ind", 33, True), synth.nof_state(0, use_quat=True, tag="bw", head_scale=0.25))
        G = 256
        ax = torch.linspace(-1.2, 1.2, G, device=dev)
        xyz = torch.stack(torch.meshgrid(ax, ax, ax, indexing="ij"), -1).reshape(-1, 3).contiguous()
        emb, nof_embs = M.Embedding(3, 10), [M.Embedding(3, 5), M.Embedding(1, 16)]
        P = xyz.shape[0]
        f_nerf, f_nof = 982528, FLOPS["nof_quat"]
        x3_nof_peak = PEAK["bf16"] * (f_nerf + f_nof) / (3 * f_nerf + 6 * f_nof)
        for tag, fn, flops, peak in (
                ("f32", lambda: M.query_sigma(xyz, nerf, emb, precision="f32"), f_nerf, PEAK["f32"]),
                ("x3", lambda: M.query_sigma(xyz, nerf, emb, precision="bf16x3"), f_nerf, PEAK["bf16"] / 3),
                ("x3_nof", lambda: M.query_sigma(xyz, nerf, emb, nof, nof_embs, 0.25, precision="bf16x3"), f_nerf + f_nof, x3_nof_peak)):
            t = timeit(fn)
            out["lattice_" + tag] = {"pts_s": sig(P / t, 4), "frac": sig(P * flops / t / 1e12 / peak)}
        del xyz
        H = W = 512
        c2w = np.array([[1, 0, 0, 0.0], [0, 1, 0, 0.0], [0, 0, 1, 4.0]], dtype=np.float64)
        rays = camera.make_rays(H, W, 1.2 * W, (W / 2, H / 2), c2w, 2.0, 6.0, -0.25)
        bg = torch.ones(H * W, 3, device=dev)
        msk = np.ones(H * W, dtype=bool)
        msk[::7] = False
        nerfs = [nerf, load(M.NeRF(8, 256, 63, [4], "ind", 5), synth.nerf_state(0, extra_feat_type="ind", extra_feat_dim=5, regime="dense", tag="fine"))]
        fn = functools.partial(M.render_rays, nerf_embeddings=[M.Embedding(3, 10), M.Embedding(1, 2), None], nerf_models=nerfs,
                               nof_embeddings=nof_embs, nof_models=[nof], N_samples=64, N_importance=128, perturb=0, noise_std=0, test_time=True)
        prev = rendering.PRECISION
        rendering.set_precision("bf16")
        try:
            t = timeit(lambda: image.render_image(rays, bg, lambda r, b: fn(r, b), 65536, msk))
        finally:
            rendering.set_precision(prev)
        nv = int(msk.sum())
        flops = nv * (64 * (f_nerf + f_nof) + 192 * (FLOPS["nerf_ind"] + f_nof))
        out["image_512_bf16"] = {"ms": sig(t * 1e3, 4), "rs_s": sig(nv * 256 / t, 4), "frac": sig(flops / t / 1e12 / PEAK["bf16"])}
    torch.cuda.empty_cache()
    try:                                         # (an auxiliary leg must not cost the driver its line)
        out["wgrad"] = wgrad_leg(torch, dev, timeit, sig)
    except Exception as e:                       # noqa: BLE001
        out["wgrad"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    torch.cuda.empty_cache()
    return out


def wgrad_leg(torch, dev, timeit, sig):
    """The training step's HBM-bound kernel against the HBM roofline (round 5): mf_weight_grads_p (three bf16 products, the
    default of the explicit backward) on the joint stage's fine-pass shapes -- the NeRF's 13 blocks over 262 144 samples
    (strided dump / gradient rows, 24.1 KB of operand rows per sample) and one NoF's 6 blocks over its 3 x 262 144
    evaluations (5.3 KB each).  ALGORITHMIC bytes = every operand row read once (4 (n_out + n_in) P per block; the heads'
    unfetched 256 columns excluded) / the launches' time, as a fraction of 8 TB/s.  Operands: ReLU-like activations
    (max(randn, 0)), gradients masked the same way -- the chip's clock answers the operands' switching activity in this kernel
    (zeros: 6.0 TB/s, randn: 4.6; profiles/r05_wgrad_x3.txt)."""
    from moco_flow_amd import autograd as A
    P, W = 262144, 256
    stride = 9 * W + W // 2
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *shape: torch.randn(*shape, device=dev, generator=g)
    acts = rn(P, stride).clamp_(min=0)
    gpre = rn(P, stride) * (rn(P, stride) > 0)
    ghead, emb64, ext32 = rn(P, 4), rn(P, 64), rn(P, 32)
    sl = lambda t, l, w=W: t[:, l * W:l * W + w]
    nerf = [(sl(gpre, l), sl(acts, l - 1), 256, 256, True) for l in range(1, 9)] + [(sl(gpre, 8), sl(acts, 7), 256, 256, True)]
    nerf += [(sl(gpre, 0), emb64, 256, 64, True), (sl(gpre, 4), emb64, 256, 64, False), (sl(gpre, 9, 128), sl(acts, 8), 128, 256, True),
             (sl(gpre, 9, 128), ext32, 128, 32, False), (ghead, acts[:, 7 * W:7 * W + 640], 4, 640, True)]
    Pn, ns = 3 * P, 4 * 128 + 16 + 16
    nacts = rn(Pn, ns).clamp_(min=0)
    ngpre = rn(Pn, ns) * (rn(Pn, ns) > 0)
    emb80 = rn(Pn, 80)
    nsl = lambda t, l, w=128: t[:, l * 128:l * 128 + w]
    nof = [(nsl(ngpre, l), nsl(nacts, l - 1), 128, 128, True) for l in (1, 2, 3)]
    nof += [(nsl(ngpre, 0), emb80, 128, 80, True), (nsl(ngpre, 2), emb80, 128, 80, False), (ngpre[:, 512:524], nsl(nacts, 3), 12, 128, True)]
    byts = lambda jobs, n: sum(4.0 * n * (a[2] + (384 if a[3] == 640 else a[3])) for a in jobs)
    out = {"precision": A.WGRAD_PRECISION, "data": "relu-like"}
    for tag, jobs, n in (("nerf13", nerf, P), ("nof6", nof, Pn)):
        t = timeit(lambda: A.weight_grads(jobs, n, dev), 5)
        out[tag] = {"ms": sig(t * 1e3, 4), "tb_s": sig(byts(jobs, n) / t / 1e12), "frac_hbm": sig(byts(jobs, n) / t / 1e12 / HBM_PEAK_TBS)}
    return out


def kernel_probe(M, rendering, torch, cfg, models, rays, bg, kw, iters=20):
    """Average duration of the DOMINANT kernel launch alone (mf_render_pass of the largest pass: the fine pass when
    there is one), HIP events on the launch stream around the C-ABI call itself -- no resample, no compaction, no
    host sync in the span.  Returns (ms, samples in that launch)."""
    from moco_flow_amd import _lib as L
    n, S = rays.shape[0], cfg["S"]
    loc = bool(cfg["nof"] in ("local", "global"))
    glob = bool(cfg["nof"] == "global")
    act = L.MF_ACT_RELU
    nofs, nof_embs = models["nofs"], models["nof_embs"]
    with torch.no_grad():
        if cfg["M"]:
            cap = {}
            M.render_rays(rays, bg, models["embs"], models["nerfs"], _capture=cap, **kw)
            z, zs, nerf, S = cap["z_fine"].contiguous(), None, models["nerfs"][1], S + cfg["M"]
        else:
            z, zs, nerf = None, torch.linspace(0, 1, S, device=rays.device), models["nerfs"][0]
        args = (rays, bg, z, zs, False, None, act, nerf, models["embs"], nofs, nof_embs, loc, glob, False, loc or glob)
        # bf16 + NoF: the per-ray bias table is prepared once, by the first warm-up call (mf_render_prepare, its own small
        # launch); the timed calls reuse it, so the events bracket the fused launch alone
        ws = [None]
        for _ in range(3):
            rendering._render_pass(*args, workspace=ws)
        # Preferred: R launches captured in ONE HIP graph and replayed -- the launches then follow each other on the device
        # without waiting for the host, so a loaded host (slow Python between two launches) cannot leak into the kernel's
        # duration; (events around the replay) / R = the kernel + the device-side launch gap.  Fallback: events around single
        # eager launches (valid as long as the host keeps the queue full).
        ms, how = None, "hip events around each eager launch (median of %d)" % iters
        try:
            R = 10
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                rendering._render_pass(*args, workspace=ws)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(R):
                    rendering._render_pass(*args, workspace=ws)
            g.replay()
            torch.cuda.synchronize()
            spans = []
            for _ in range(max(3, iters // R + 1)):
                s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s0.record()
                g.replay()
                e0.record()
                torch.cuda.synchronize()
                spans.append(s0.elapsed_time(e0) / R)
            ms = float(sorted(spans)[len(spans) // 2])
            how = f"hip events around a HIP-graph replay of {R} back-to-back launches, / {R} (median of {len(spans)} replays)"
            del g
        except Exception as exc:  # noqa: BLE001  (graph capture unavailable: time eager launches)
            how += f" [graph probe unavailable: {type(exc).__name__}]"
        if ms is None:
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
            for s, e in ev:
                s.record()
                rendering._render_pass(*args, workspace=ws)
                e.record()
            torch.cuda.synchronize()
            t = sorted(s.elapsed_time(e) for s, e in ev)
            ms = float(t[len(t) // 2])
    return ms, n * S, how


def run_config(name, a, ctx, steps, warmup, main):
    """Time `steps` steps of configuration `name` on this rank.  Returns the result dict (rank-local timing
    already reduced with MAX over ranks)."""
    import numpy as np
    import torch
    M, synth, rendering, dist, dev, rank, world = (ctx[k] for k in ("M", "synth", "rendering", "dist", "dev", "rank", "world"))
    cfg = CONFIGS[name]
    rendering.set_precision(cfg["precision"])
    models = build_models(M, synth, dev, cfg)
    n = cfg["rays"]
    rays_np, bg_np = synth.rays(0, n * world, chained=(cfg["nof"] == "global"))    # weak scaling: contiguous blocks
    lo = rank * n
    rays = torch.from_numpy(rays_np[lo:lo + n]).to(dev)
    bg = torch.from_numpy(bg_np[lo:lo + n]).to(dev)
    gt = torch.from_numpy(synth.uniform01(123 + rank, n * 3).reshape(n, 3).astype(np.float32)).to(dev)
    kw = render_kwargs(cfg, models)

    from moco_flow_amd.dist import N_PARTIALS, OverlappedLossReducer
    # the loss partials + their all-reduce belong to the configs that name them (C4, C5): the C2 main line is the same work
    # at every N, so the driver's 1 -> N curve compares like with like
    with_loss = bool(cfg.get("loss"))
    reducer = OverlappedLossReducer(N_PARTIALS, dev) if with_loss else None
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]

    def step(i=None):
        if i is not None:
            ev[i][0].record()
        # with a loss: the fast path of the mean-only caller -- mf_loss_partials instead of mask compaction +
        # host sync; the 12 partials (96 B) go to the asynchronous all-reduce (RCCL over xGMI when world > 1)
        out = M.render_rays(rays, bg, models["embs"], models["nerfs"], _loss_target=gt if reducer is not None else None, **kw)
        # what the unchanged trainer does with the consensus vectors right away (trainer_moco_flow.py:317-328): their means
        # (lazy.MaskedVector: masked sums on the device, no compaction, no host sync)
        # -- behind the per-chunk concatenation of its forward() wrapper, `results[k] = torch.cat(v, 0)` (:199-223; one chunk:
        # chunk >= N_rand in every shipped YAML), which stays lazy too
        cons = [torch.mean(torch.cat([v], 0)) for k, v in out.items() if k.startswith("nof_")]
        if i is not None:
            ev[i][1].record()
        if reducer is not None:
            reducer.push(out["loss_partials"], donate=True)      # (nothing reads the local partials afterwards)
        return out

    # everything built so far (torch's import, the models) leaves the collector's generations: the later collections of
    # every leg scan the few objects of a step, not the whole heap
    gc.collect()
    gc.freeze()
    with torch.no_grad():
        if os.environ.get("MF_BENCH_STEP_TIMES") == "1" and rank == 0:
            # diagnosis: host time of each of the first calls, synchronised one by one (stderr)
            ts = []
            for _ in range(12):
                t1 = time.perf_counter()
                out = step()
                torch.cuda.synchronize()
                ts.append(round((time.perf_counter() - t1) * 1e3, 3))
            print("first calls, ms each (synchronised):", ts, file=sys.stderr)
        for _ in range(warmup):
            out = step()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        # the interpreter's cyclic collector stays out of the timed region (as `timeit` does): a full collection of the heap
        # torch's import leaves behind blocks the host for 40-50 ms once, at a call of its own choosing -- in a 20-step
        # region of a 0.34 ms pass that is the difference between 0.37 and 2.3 ms per step (MF_BENCH_STEP_TIMES=2 shows it)
        gc_was = gc.isenabled()
        if os.environ.get("MF_BENCH_GC") != "1":
            gc.collect()
            gc.disable()
        t0 = time.perf_counter()
        host_ts = [] if os.environ.get("MF_BENCH_STEP_TIMES") == "2" else None
        for _ in range(steps):                  # the timed region: exactly `steps` calls, nothing else on the stream
            out = step()
            if host_ts is not None:
                host_ts.append(time.perf_counter())
        if reducer is not None:
            reducer.finish()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if gc_was:
            gc.enable()
        if host_ts and rank == 0:
            print("host time of each timed call's enqueue, ms:", [round((b - a) * 1e3, 3) for a, b in zip([t0] + host_ts, host_ts)],
                  "then the final synchronize: %.3f" % ((t0 + elapsed - host_ts[-1]) * 1e3), file=sys.stderr)
        # per-call device span (step_span_ms), outside the timed region: the event pairs cost ~10 us of stream time each on
        # this runtime, which is 3 % of a 0.33 ms step
        for i in range(steps):
            out = step(i)
        if reducer is not None:
            reducer.finish()
        torch.cuda.synchronize()
        if os.environ.get("MF_BENCH_TRACE_OPS") == "1" and rank == 0:
            # which torch ops (copies, fills, allocations' memsets) ride along with the HIP launches of one step: stderr
            from torch.profiler import ProfilerActivity, profile
            with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
                step()
                if reducer is not None:
                    reducer.finish()
                torch.cuda.synchronize()
            print(prof.key_averages(group_by_stack_n=6).table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60),
                  file=sys.stderr)
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernel_ms = float(np.median([s.elapsed_time(e) for s, e in ev]))        # median of the timed calls

    spr = samples_per_ray(cfg)
    value = n * spr * world * steps / elapsed
    flops_step = n * spr * flops_per_sample(cfg)
    step_span_ms = kernel_ms
    kernel_ms, launch_samples, probe_how = kernel_probe(M, rendering, torch, cfg, models, rays, bg, kw)
    flops_launch = launch_samples * flops_per_sample(cfg)
    achieved = flops_launch / (kernel_ms * 1e-3) / 1e12
    peak = peak_of(cfg)
    traffic, traffic_src, prof = traffic_of(cfg.get("profile", name))
    alg_bytes = algorithmic_bytes(cfg, launch_samples)
    launches = 2 if cfg["M"] else 1
    # three readings of the same fraction, so that they are never confused:
    #   frac             HIP-graph replay of back-to-back launches of the dominant kernel, median (warm box: the best case)
    #   frac_step        the whole step's algorithmic FLOP / ms_per_step of the timed region (everything a step does)
    #   frac_rocprof_avg the rocprofv3 --kernel-trace average of the same command in profiles/ (+ MFMA busy and the effective
    #                    clock of that run: busy x GHz / 2.4 / (issued / algorithmic MFMAs) = the fraction)
    # everything below comes from the COMMITTED profiles of this same command (profiles/traffic.json, written by
    # tools/summarize_prof.py) and does not move with this run: kept apart under "from_profiles" 
    rp = {"source": traffic_src, "traffic": traffic, "traffic_ratio": (traffic / alg_bytes) if traffic else None}
    if prof.get("rocprof_avg_us"):
        # (with a fine pass the trace's average runs over coarse AND fine dispatches of the same kernel, one of each per step:
        #  the step's algorithmic FLOP over launches x average)
        # (by_pass: tools/summarize_prof.py's split of the dominant kernel's dispatches into the coarse and the fine pass' launches)
        bp = prof.get("by_pass")
        step_us = (bp["coarse"]["avg_us"] + bp["fine"]["avg_us"]) if (bp and launches == 2) else launches * prof["rocprof_avg_us"]
        rp.update({"frac_rocprof_avg": flops_step / (step_us * 1e-6) / 1e12 / peak, "rocprof_avg_us": prof["rocprof_avg_us"],
                   "mfma_busy": prof.get("mfma_busy"), "ghz": prof.get("ghz"), "profile": prof.get("profile")})
    res = {
        "value": value, "ms_per_step": elapsed / steps * 1e3, "dtype": cfg["precision"],
        "config": {"workload": cfg["what"] + f" [{cfg['precision']}]", "rays_per_gpu": n,
                   "samples_per_ray": spr, "global_rays": n * world,
                   "sharding": f"rays{world}" if world > 1 else "none",
                   "loss_allreduce": bool(reducer is not None and world > 1),
                   #  the C2 main line carries no loss path and no collective at any N -- its 1 -> N curve is N
                   # independent launches by design; the configs that name a collective (C4, C5) are the legs
                   "main_has_collective": bool(reducer is not None and world > 1)},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "peak_note": PEAK_NOTE[cfg["precision"]],
                     "unit": "TFLOP/s", "frac": achieved / peak,
                     "frac_step": flops_step / (elapsed / steps) / 1e12 / peak,
                     "traffic": traffic, "traffic_unit": "B/launch", "traffic_algorithmic": alg_bytes,
                     "from_profiles": rp,
                     "traffic_note": "HBM bytes per launch: the weight stream is read once per XCD L2 (8 x the packed weights) + "
                                     "rays / requested (N,S) planes" + (" + the per-ray NoF bias table" if cfg["nof"] and cfg["precision"] != "f32" else "")
                                     + "; algorithmic I/O is 68 B/ray (+ 8 B/sample per requested plane): MFMA-bound, not HBM-bound",
                     "kernel": "mf_render_pass" + (" (fine pass)" if cfg["M"] else ""),
                     "kernel_ms": kernel_ms, "kernel_ms_how": probe_how, "flops_per_launch": flops_launch, "samples_per_launch": launch_samples,
                     "step_span_ms": step_span_ms, "launches_per_step": launches,
                     "flops_per_step": flops_step},
    }
    if cfg["nof"] and reducer is None and world == 1:
        # the same pass for a mean-only caller (the trainer's loss terms): consensus sums from mf_loss_partials instead of
        # the compacted data-dependent-length vectors -- no host sync, so consecutive steps pipeline
        with torch.no_grad():
            for _ in range(warmup):
                M.render_rays(rays, bg, models["embs"], models["nerfs"], _loss_target=gt, **kw)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(steps):
                M.render_rays(rays, bg, models["embs"], models["nerfs"], _loss_target=gt, **kw)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t1) / steps * 1e3
        res["loss_path"] = {"ms_per_step": ms, "value": n * spr / (ms * 1e-3), "unit": "ray-samples/s",
                            "what": "render_rays(_loss_target=gt): the 12 loss partials instead of the masked consensus "
                                    "vectors (no compaction, no host sync; INTEGRATION.md)"}
    if main and rank == 0 and world == 1 and not a.no_train_leg and cfg["precision"] == "f32":
        res["fwd_bwd"] = train_leg(M, torch, models, rays, bg, gt, kw, cfg)
        if not a.no_extra_legs:
            res["train_steps"] = train_shape_legs(M, synth, torch, rays.device)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        if main:
            base, ref = cpu_baseline(cfg, models["states"], rays_np[:n], bg_np[:n])
            res["cpu_baseline"] = base
            res["speedup_vs_cpu"] = value / base["value"]
            if name == "C2":
                # BASELINE config C1: the same canonical NeRF at 1024 rays x 64 samples on the host cores
                ts = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    oracle_render(cfg, models["states"], torch.from_numpy(rays_np[:1024]), torch.from_numpy(bg_np[:1024]))
                    ts.append(time.perf_counter() - t0)
                med = float(np.median(ts))
                res["c1_cpu"] = dict(value=1024 * spr / med, unit="ray-samples/s", cores=base["cores"], kind="port",
                                     sample=f"BASELINE config C1: canonical NeRF, 1024 rays x {spr} samples, CPU oracle, median of 3 "
                                            f"({med:.3f} s), {base['cores']} threads on {cpu_model()}")
        k = n if (main and not cfg["M"]) else min(512, n)
        if not (main and not cfg["M"]):             # accuracy on a bounded 512-ray sample; with a fine pass the
            cap = {}                                # oracle evaluates it on the HIP path's own fine depths
            with torch.no_grad():
                out = M.render_rays(rays[:k], bg[:k], models["embs"], models["nerfs"], _capture=cap, **kw)
            t0 = time.perf_counter()
            ref = oracle_render(cfg, models["states"], torch.from_numpy(rays_np[:k]), torch.from_numpy(bg_np[:k]),
                                z_fine=cap["z_fine"].cpu() if cfg["M"] else None)
            dt = time.perf_counter() - t0
            if not main:      # every leg carries its own baseline: the oracle on this bounded sample of the leg's workload
                res["cpu_baseline"] = dict(value=k * spr / dt, unit="ray-samples/s", cores=torch.get_num_threads(), kind="port",
                                           sample=f"one oracle pass over the first {k} rays x {spr} samples of the batch ({dt:.2f} s), "
                                                  f"{torch.get_num_threads()} threads on {cpu_model()}")
                res["speedup_vs_cpu"] = value / res["cpu_baseline"]["value"]
        res["error_vs_cpu"] = errors_vs(ref, out)
        res["error_vs_cpu"]["sample"] = f"first {k} rays of the batch vs the CPU oracle" + (
            " (fine pass on the HIP path's own resampled depths)" if cfg["M"] else "")
    return res


COMPACT_LIMIT = 1800        # bytes: the driver's record keeps the final 2000 characters of stdout
LEG_FIELDS = ["ms", "frac", "frac_step", "db", "max_rel"]


def sig(x, n=4):
    """x to n significant digits (None / non-finite -> None: the compact line must always json.loads strictly)."""
    if x is None:
        return None
    x = float(x)
    return float(f"{x:.{n}g}") if x == x and abs(x) != float("inf") else None


def compact_leg(r):
    """One measured configuration as LEG_FIELDS: ms per step, roofline fraction of the dominant kernel (graph replay), roofline
    fraction of the whole step, PSNR-equivalent dB and worst max-rel of the per-ray outputs against the CPU oracle."""
    e = r.get("error_vs_cpu") or {}
    worst = max(e["max_rel"].values()) if e.get("max_rel") else None
    rf = r["roofline"]
    return [sig(r["ms_per_step"]), sig(rf["frac"], 3), sig(rf["frac_step"], 3), sig(e.get("psnr_equiv_db")), sig(worst, 2)]


def assemble(a, world, main_cfg, res, leg_results, extras):
    """-> (detail, compact).  `detail` is everything measured (tens of KB: written to gpurun_out/bench_detail_n<N>.json and to
    stderr); `compact` is the ONE stdout line the driver parses: the contract keys + roofline + cpu_baseline + one array per
    leg, numbers to 4 significant digits, no prose -- at most COMPACT_LIMIT bytes (tests/test_host_cpu.py holds it to that)."""
    head = {"metric": "ray-samples/sec (4096 rays x 64 samples)", "value": res["value"], "unit": "ray-samples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": res["ms_per_step"],
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": res["dtype"], "data": "synthetic"}
    detail = dict(head, config=res["config"], roofline=res["roofline"])
    for k in ("fwd_bwd", "train_steps", "cpu_baseline", "c1_cpu", "error_vs_cpu", "speedup_vs_cpu", "loss_path", "rccl_ranks_seen"):
        if k in res:
            detail[k] = res[k]
    if leg_results:
        detail["configs"] = leg_results
    detail.update(extras)

    rf, cfg = res["roofline"], res["config"]
    fp = rf.get("from_profiles") or {}
    compact = dict(head, value=sig(res["value"], 6), ms_per_step=sig(res["ms_per_step"], 5))
    compact["config"] = {"workload": CONFIGS[main_cfg].get("short", main_cfg),
                         **{k: cfg[k] for k in ("rays_per_gpu", "samples_per_ray", "global_rays", "sharding", "loss_allreduce", "main_has_collective")}}
    compact["roofline"] = {"bound": rf["bound"], "achieved": sig(rf["achieved"]), "peak": rf["peak"], "unit": rf["unit"],
                           "frac": sig(rf["frac"]), "frac_step": sig(rf["frac_step"]), "traffic": None if rf["traffic"] is None else int(sig(rf["traffic"])),
                           "traffic_algorithmic": rf["traffic_algorithmic"], "kernel": rf["kernel"], "kernel_ms": sig(rf["kernel_ms"]),
                           "from_profiles": {"frac_rocprof_avg": sig(fp.get("frac_rocprof_avg")), "mfma_busy": sig(fp.get("mfma_busy")),
                                             "ghz": sig(fp.get("ghz")),
                                             "profile": (fp.get("profile") or "").replace("profiles/", "").replace("_summary.txt", "") or None}}
    if "cpu_baseline" in res:
        cb = res["cpu_baseline"]
        compact["cpu_baseline"] = {"value": sig(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                   "sample": cb.get("sample_short", cb["sample"])[:60], "cpu": cb.get("cpu", cpu_model())[:40]}
        compact["speedup_vs_cpu"] = sig(res.get("speedup_vs_cpu"))
    if "rccl_ranks_seen" in res:
        compact["rccl_ranks_seen"] = res["rccl_ranks_seen"]
    e = res.get("error_vs_cpu") or {}
    if e.get("max_rel"):
        worst = max(e["max_rel"], key=e["max_rel"].get)
        compact["error_vs_cpu"] = {"max_rel": sig(e["max_rel"][worst], 3), "key": worst, "db": sig(e.get("psnr_equiv_db"))}
    compact["leg_fields"] = LEG_FIELDS
    legs = {} if leg_results else {main_cfg: compact_leg(res)}     # (the main line's own numbers are the top-level keys)
    for name, r in leg_results.items():
        legs[name] = compact_leg(r)
    compact["legs"] = legs
    if "train_steps" in res:
        short = {"stage1": "s1", "joint": "joint", "joint_with_optimizer": "joint_adam"}        # (the opt-in forwards: detail only)
        ts = {short[k]: v for k, v in res["train_steps"].items() if k in short}
        compact["train_ms"] = {k: sig(v["ms_per_step"]) for k, v in ts.items()}
        gb = {k: sig(v["hbm_gb_per_step"], 3) for k, v in ts.items() if v.get("hbm_gb_per_step") and k in ("s1", "joint")}
        if gb:
            compact["train_hbm_gb"] = gb
    if "fwd_bwd" in res:
        compact["fwd_bwd_ms"] = sig(res["fwd_bwd"]["ms_per_step"])
    if "train_joint_dp" in extras:
        dp = extras["train_joint_dp"]
        compact["train_dp_ms"] = {"local": sig(dp["ms_per_step_local"]), "allreduce": sig(dp["ms_per_step_allreduce"])}
    if "aux" in extras:
        # roofline fraction per auxiliary path (lattice sigma query, 512^2 image: of the matrix peak; wgrad: of the HBM peak)
        ax, c = extras["aux"], {}
        for k, v in ax.items():
            if k.startswith("lattice_"):
                c[k.replace("lattice", "lat")] = v["frac"]
            elif k.startswith("image_"):
                c["img512"] = v["frac"]
            elif k == "wgrad" and "nerf13" in v:
                c["wg_nerf"], c["wg_nof"] = v["nerf13"]["frac_hbm"], v["nof6"]["frac_hbm"]
        compact["aux_frac"] = c
    return detail, compact


def emit(detail, compact, world):
    """Detail first (side file + ONE stderr line that does not start with '{'), then the compact line LAST on stdout."""
    blob = json.dumps(detail)
    try:
        if not compact.get("dryrun"):            # (a dry run leaves the real runs' side files alone)
            out = os.path.join(ROOT, "gpurun_out")
            os.makedirs(out, exist_ok=True)
            with open(os.path.join(out, "bench_detail_n%d.json" % world), "w") as fh:
                fh.write(blob + "\n")
    except OSError as exc:
        print(f"bench_detail: could not write the side file ({exc})", file=sys.stderr)
    print("bench_detail " + blob, file=sys.stderr, flush=True)
    # never out-grow the reporting channel again: shed the optional tails (never the contract keys) until the line fits
    for k in ("aux_frac", "train_hbm_gb", "train_dp_ms", "fwd_bwd_ms", "train_ms", "error_vs_cpu"):
        if len(json.dumps(compact)) <= COMPACT_LIMIT:
            break
        compact.pop(k, None)
    while len(json.dumps(compact)) > COMPACT_LIMIT and len(compact.get("legs", {})) > 1:
        compact["legs"].popitem()
    line = json.dumps(compact, allow_nan=False)
    assert len(line) <= COMPACT_LIMIT, len(line)
    sys.stdout.flush()
    print(line, flush=True)


def leg_names(world):
    """The short extra legs of a default run: every other BASELINE config at N = 1; at N > 1 the headline WITHOUT its
    collective (C2: what the C2dp main line adds the loss all-reduce to) + the configs that name a collective (C4, C5)."""
    return ["C2x", "C3", "C3x", "C3g", "C5", "C5x", "C5full", "C5xfull"] if world == 1 else ["C2", "C4", "C5"]


def main_config(a, world):
    return "C2dp" if (a.config == "C2" and world > 1) else a.config


def fake_result(name, world, seed):
    """MF_BENCH_DRYRUN: a result shaped exactly like run_config()'s, with full-length floats (worst case for the line length)."""
    cfg = CONFIGS[name]
    x = 0.123456789012345 + 0.01 * seed
    with_loss = bool(cfg.get("loss"))
    prof = traffic_of(cfg.get("profile", name))[2]
    res = {"value": 118962552.06278363 * world * (1 + x), "ms_per_step": 2.2035841990145855 * (1 + x), "dtype": cfg["precision"],
           "config": {"workload": cfg["what"] + f" [{cfg['precision']}]", "rays_per_gpu": cfg["rays"], "samples_per_ray": samples_per_ray(cfg),
                      "global_rays": cfg["rays"] * world, "sharding": f"rays{world}" if world > 1 else "none",
                      "loss_allreduce": with_loss and world > 1, "main_has_collective": with_loss and world > 1},
           "roofline": {"bound": "mfma", "achieved": 139.76543219876 * (1 + x), "peak": peak_of(cfg), "peak_note": PEAK_NOTE[cfg["precision"]],
                        "unit": "TFLOP/s", "frac": 0.88912345678 + 0.001 * seed, "frac_step": 0.8991234567 + 0.001 * seed,
                        "traffic": 20912345.678, "traffic_unit": "B/launch", "traffic_algorithmic": algorithmic_bytes(cfg, cfg["rays"] * cfg["S"]),
                        "from_profiles": {"source": "profiles/traffic.json", "traffic": 20912345.678, "traffic_ratio": 7.87654321,
                                          "frac_rocprof_avg": 0.8887654321, "rocprof_avg_us": 2225.8123, "mfma_busy": 0.9112345,
                                          "ghz": 2.3512345, "profile": prof.get("profile", "profiles/r06_c5xfull_summary.txt")},
                        "kernel": "mf_render_pass" + (" (fine pass)" if cfg["M"] else ""), "kernel_ms": 2.15312345678 * (1 + x),
                        "kernel_ms_how": "dryrun", "flops_per_launch": 311117381632, "samples_per_launch": 262144,
                        "step_span_ms": 2.21234567, "launches_per_step": 2 if cfg["M"] else 1, "flops_per_step": 311117381632},
           "error_vs_cpu": {"max_rel": {"rgb_coarse": 4.412345678e-07 * (1 + seed), "depth_coarse": 3.912345678e-07, "opacity_coarse": 2.1e-07},
                            "l2_rel": {"rgb_coarse": 1.2e-07}, "psnr_equiv_db": 140.12345678, "psnr_key": "rgb_coarse", "sample": "dryrun"},
           "cpu_baseline": {"value": 125842.77898723527, "unit": "ray-samples/s", "cores": 32, "kind": "port", "cpu": cpu_model(),
                            "sample": "4 full 4096x64 batches at the fastest of a thread sweep (16t 2.51s, 32t 2.08s, 64t 2.31s), median 2.083 s/batch",
                            "sample_short": "4 full batches, 16/32/64t sweep"},
           "speedup_vs_cpu": 945.3123456789}
    return res


def dryrun_worker(a, rank, world):
    """MF_BENCH_DRYRUN=1 (CPU control-flow test of the self-spawn / rendezvous / reduction / REPORTING path, no GPU):
    gloo process group, the ranks-seen all-reduce, the overlapped loss reducer on CPU tensors, MAX-over-ranks timing, then the
    same assemble() / emit() as a real run over results shaped like run_config()'s with full-length floats -- so the CPU
    suite holds the compact stdout line to its byte limit and its required keys for every leg set."""
    import torch
    import torch.distributed as dist
    from moco_flow_amd.dist import N_PARTIALS, OverlappedLossReducer
    if world > 1:
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    seen = ranks_seen(torch, dist if world > 1 else None, "cpu")
    red = OverlappedLossReducer(N_PARTIALS, "cpu")
    t0 = time.perf_counter()
    for i in range(a.steps):
        red.push(torch.full((N_PARTIALS,), float(rank + 1), dtype=torch.float64))
    tot = red.finish()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
    if rank == 0:
        main_cfg = main_config(a, world)
        res = fake_result(main_cfg, world, 0)
        res["rccl_ranks_seen"] = seen
        legs, extras = {}, {}
        if not a.no_extra_legs and a.config == "C2":
            legs = {n: fake_result(n, world, i + 1) for i, n in enumerate(leg_names(world))}
            if world == 1:
                ts = lambda ms: {"ms_per_step": ms * 1.0123456789, "rays": 1024, "samples_per_ray": 384, "hbm_gb_per_step": 41.0123456}
                res["train_steps"] = {"what": "dryrun", "stage1": ts(36.09), "stage1_optin_bf16x3_forward": ts(27.75), "joint": ts(15.56),
                                      "joint_with_optimizer": ts(15.82), "joint_optin_bf16x3_forward": ts(13.42)}
                res["fwd_bwd"] = {"ms_per_step": 10.123456789}
                extras["aux"] = {"lattice_f32": {"pts_s": 140100000.0, "frac": 0.875}, "lattice_x3": {"pts_s": 491100000.0, "frac": 0.575},
                                 "lattice_x3_nof": {"pts_s": 413300000.0, "frac": 0.617}, "image_512_bf16": {"ms": 56.96, "rs_s": 1010000000.0, "frac": 0.508},
                                 "wgrad": {"precision": "bf16x3", "data": "relu-like", "nerf13": {"ms": 1.49, "tb_s": 4.35, "frac_hbm": 0.544},
                                           "nof6": {"ms": 1.012, "tb_s": 4.12, "frac_hbm": 0.515}}}
            else:
                extras["train_joint_dp"] = {"ms_per_step_local": 15.5612345, "ms_per_step_allreduce": 15.9812345, "world": world, "buckets": 4,
                                            "flat_gradient_bytes": 5300000}
        detail, compact = assemble(a, world, main_cfg, res, legs, extras)
        compact["dryrun"] = [tot[-1].tolist()[0], sig(float(t.item()), 2)]            # [reduced partial, elapsed s]
    if world > 1:
        flush_c_stdio()
        dist.barrier()
        dist.destroy_process_group()
    flush_c_stdio()
    if rank == 0:
        emit(detail, compact, world)


def flush_c_stdio():
    """RCCL prints a version banner (five lines) through C stdio to STDOUT when its first communicator comes up; with stdout a pipe
    that text sits in libc's buffer until the process exits -- i.e. it would land BEHIND the compact JSON line, and a driver that
    parses the last stdout line would find "Librccl path : ..." there (seen in tests/rccl_child.py's log).  Flushing libc's
    streams right after the communicator exists (every rank) and again before rank 0 prints keeps the JSON line last."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001  (no libc handle: nothing buffered through it either)
        pass
    sys.stdout.flush()


def ranks_seen(torch, dist, dev):
    """How many ranks the process group's all-reduce actually reached: a ones-vector summed through the backend the run uses
    ("nccl" = RCCL on the GPU box) BEFORE the timed region -- on the compact line as `rccl_ranks_seen`, so a scaling record
    proves the collective saw N ranks.  1 without a process group."""
    if dist is None:
        return 1
    ones = torch.ones(8, device=dev, dtype=torch.float32)
    dist.all_reduce(ones, op=dist.ReduceOp.SUM)
    v = ones.cpu().tolist()
    assert all(x == v[0] for x in v), v
    flush_c_stdio()                         # (the communicator exists now: RCCL's banner leaves libc's buffer here, not at exit)
    return int(v[0])


def worker(a):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if os.environ.get("MF_BENCH_DRYRUN"):
        return dryrun_worker(a, rank, world)
    import torch
    if os.environ.get("MF_BENCH_SHARE_GPU"):                          # control-flow tests: every rank on GPU 0
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # (MF_BENCH_FORCE_DIST=1: a process group at world size 1 too -- the GPU test of the reporting channel under a REAL RCCL communicator)
    if world > 1 or os.environ.get("MF_BENCH_FORCE_DIST"):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("MF_BENCH_BACKEND", "nccl")           # "nccl" IS RCCL on ROCm; "gloo": control-flow tests only
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    import moco_flow_amd as M
    from moco_flow_amd import rendering, synth
    rendering.STRICT_RNG = False        # noise_std = 0: do not launch the reference's dead randn (rendering.py:166)
    M._lib.lib()                        # fail loudly if the HIP library is missing
    ctx = dict(M=M, synth=synth, rendering=rendering, dist=dist, dev=dev, rank=rank, world=world)
    seen = ranks_seen(torch, dist, dev)

    main_cfg = main_config(a, world)
    res = run_config(main_cfg, a, ctx, a.steps, a.warmup, main=True)
    res["rccl_ranks_seen"] = seen
    leg_results, extras = {}, {}
    if not a.no_extra_legs and a.config == "C2":
        for name in leg_names(world):
            # sub-millisecond passes: 100 warm-up steps (a handful leaves the clocks ramping -- C3 measured 0.372 ms/step
            # after 5 warm-up steps, 0.345 after 100, kernel 0.346 -- profiles/README.md) and 200 timed ones
            k_leg, w_leg = CONFIGS[name].get("steps", (200, 100)) if a.steps >= 10 else (a.steps, a.warmup)
            r = run_config(name, a, ctx, k_leg, w_leg, main=False)
            leg_results[name] = {k: r[k] for k in ("value", "ms_per_step", "dtype", "config", "roofline", "error_vs_cpu", "loss_path",
                                                   "cpu_baseline", "speedup_vs_cpu") if k in r}
            leg_results[name].update(steps=k_leg, warmup=w_leg)
    if not a.no_extra_legs and not a.no_train_leg and a.config == "C2":
        if world > 1:
            extras["train_joint_dp"] = train_dp_leg(M, synth, torch, dev, dist, rank, world)
        else:
            extras["aux"] = aux_legs(M, synth, torch, dev)
    # the compact line must be the LAST thing on stdout: every rank empties libc's buffers, all meet, the group goes down (whatever
    # that prints comes first), and only then rank 0 prints
    if dist is not None:
        flush_c_stdio()
        dist.barrier()
        dist.destroy_process_group()
    flush_c_stdio()
    if rank == 0:
        detail, compact = assemble(a, world, main_cfg, res, leg_results, extras)
        emit(detail, compact, world)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="C2",
                    help="BASELINE.json configuration of the main line (default C2 = the headline, at every N)")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the short legs of the other BASELINE configs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-leg", action="store_true",
                    help="skip the extra fwd+bwd measurement (SURVEY.md §8d: reported separately from the graded forward)")
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 and world == 1 and not os.environ.get("MF_BENCH_CHILD"):
        sys.exit(spawn_workers(a.gpus, sys.argv[1:]))      # nothing above touched the GPU (torch is not even imported)
    a.gpus = world
    worker(a)


if __name__ == "__main__":
    main()
