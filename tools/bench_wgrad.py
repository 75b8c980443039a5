#!/usr/bin/env python3
"""Micro-benchmark of mf_weight_grads (and mf_nerf_backward) on the stage-1 fine-pass shape
(5120 rays x 256 samples): per-shape item sets, to calibrate the scheduler's cost model and to track
the kernels' MFMA efficiency.  Usage: bench_wgrad.py [P]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import moco_flow_amd as M
from moco_flow_amd import autograd as A

P = int(sys.argv[1]) if len(sys.argv) > 1 else 5120 * 256
# MF_WGRAD_DATA: "randn" (default) | "zero" | "relu" (activations max(randn, 0), gradients randn masked likewise): what the operands'
# bit patterns cost -- the chip answers switching activity with its clock (profiles/r05_duty_power.txt)
_kind = os.environ.get("MF_WGRAD_DATA", "randn")
_randn = torch.randn
if _kind == "zero":
    torch.randn = lambda *a, **k: torch.zeros(*a, **k)
elif _kind == "relu":
    torch.randn = lambda *a, **k: _randn(*a, **k).clamp_(min=0)
dev = torch.device("cuda")
D, W = 8, 256
stride = (D + 1) * W + W // 2
acts = torch.randn(P, stride, device=dev)
gpre = torch.randn((P + 127) // 128 * 128, stride, device=dev)[:P]
ghead = torch.randn(P, 4, device=dev)
emb64 = torch.randn(P, 64, device=dev)
ext32 = torch.randn(P, 32, device=dev)
sl = lambda t, l, w=W: t[:, l * W:l * W + w]


def timeit(f, n=5):
    f(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n * 1e3


sets = {
    "A x9 (256x256)": [(sl(gpre, l), sl(acts, l - 1), 256, 256, True) for l in range(1, 9)] + [(sl(gpre, 8), sl(acts, 7), 256, 256, True)],
    "A x1": [(sl(gpre, 1), sl(acts, 0), 256, 256, True)],
    "B x2 (256x64)": [(sl(gpre, 0), emb64, 256, 64, True), (sl(gpre, 4), emb64, 256, 64, False)],
    "C x1 (128x256)": [(sl(gpre, 9, 128), sl(acts, 8), 128, 256, True)],
    "D x1 (128x32)": [(sl(gpre, 9, 128), ext32, 128, 32, False)],
    "E x1 (4x640)": [(ghead, acts[:, 7 * W:7 * W + 640], 4, 640, True)],
}
# the same nine blocks with every layer's rows contiguous ((layer, P, 256) planes instead of (P, layers x 256) rows): what a
# plane-major dump would give the reader
if os.environ.get("MF_WGRAD_PLANES") == "1":
    gpl = torch.randn(9, P, W, device=dev)
    apl = torch.randn(9, P, W, device=dev)
    sets["A x9 planes"] = [(gpl[l], apl[l - 1], 256, 256, True) for l in range(1, 9)] + [(gpl[8], apl[7], 256, 256, True)]
# NoF shapes (one evaluation: D = 4, W = 128, skip at 2; dump stride 4*128 + 16, embedded input 80 columns)
nacts = torch.randn(P, 4 * 128 + 16, device=dev)
ngpre = torch.randn((P + 127) // 128 * 128, 4 * 128 + 16, device=dev)[:P]
emb80 = torch.randn(P, 80, device=dev)
nsl = lambda t, l, w=128: t[:, l * 128:l * 128 + w]
sets["F x3 (128x128)"] = [(nsl(ngpre, l), nsl(nacts, l - 1), 128, 128, True) for l in (1, 2, 3)]
sets["F x1"] = sets["F x3 (128x128)"][:1]
sets["G x2 (128x80)"] = [(nsl(ngpre, 0), emb80, 128, 80, True), (nsl(ngpre, 2), emb80, 128, 80, False)]
sets["H x1 (12x128)"] = [(ngpre[:, 512:524], nsl(nacts, 3), 12, 128, True)]
sets["NoF all 6"] = sets["F x3 (128x128)"] + sets["G x2 (128x80)"] + sets["H x1 (12x128)"]
sets_check = None
A9, B2, C1, D1 = (sets[k] for k in ("A x9 (256x256)", "B x2 (256x64)", "C x1 (128x256)", "D x1 (128x32)"))
if os.environ.get("MF_WGRAD_MIX") == "1":          # how the cost model balances mixed launches (the sum of the parts is the target)
    sets["mix A9+C"] = A9 + C1
    sets["mix A9+B2"] = A9 + B2
    sets["mix A9+D"] = A9 + D1
    sets["mix A9+B2+C+D"] = A9 + B2 + C1 + D1
    sets["mix B2+C+D"] = B2 + C1 + D1
sets["all 13"] = sum((sets[k] for k in ("A x9 (256x256)", "B x2 (256x64)", "C x1 (128x256)", "D x1 (128x32)", "E x1 (4x640)")), [])
flops = lambda jobs: sum(2.0 * P * a[2] * a[3] for a in jobs)
byts = lambda jobs: sum(4.0 * P * (a[2] + a[3]) for a in jobs)
prec = os.environ.get("MF_WGRAD", "f32")          # MF_WGRAD=bf16x3: the three-product variants of the large blocks
A.set_wgrad_precision(prec)
print(f"P = {P} samples, wgrad precision {prec}")
only = os.environ.get("MF_WGRAD_SETS")            # comma-separated prefixes of the set names to time (timing-ablation runs); no check then
for k, jobs in sets.items():
    if only and not any(k.startswith(o) for o in only.split(",")):
        continue
    ms = timeit(lambda: A.weight_grads(jobs, P, dev))
    print(f"  {k:18s}: {ms:7.3f} ms  {flops(jobs)/ms/1e9:7.1f} TFLOP/s  {byts(jobs)/ms/1e9:6.2f} TB/s"
          f"  ({ms/len(jobs)*1e6/((P+15)//16)*256*2.4/1e3:7.0f} CU-cycles/stage/item @2.4GHz)")
if only:
    sys.exit(0)
# correctness spot check against library GEMMs
jobs = sets["all 13"] + sets["NoF all 6"]
res = A.weight_grads(jobs[:13], P, dev) + A.weight_grads(jobs[13:], P, dev)
worst = 0.0
for (G, X, no, ni, b), (dW, db) in zip(jobs, res):
    ref = (G.t().double() @ X.double())
    if ni == 640:
        ref[:, 256:512] = 0      # heads block: the `final` columns are not fetched (dW = 0 there by contract)
    err = float((dW[:no].double() - ref).abs().max() / ref.abs().max())
    print(f"    {no}x{ni}: max-rel {err:.2e}  l2-rel {float((dW[:no].double() - ref).norm() / ref.norm()):.2e}")
    worst = max(worst, err)
    if b:
        rb = G.sum(0)
        worst = max(worst, float((db[:no] - rb).abs().max() / rb.abs().max()))
print(f"  max-rel vs library GEMM: {worst:.2e}")
