"""Which torch-side device work is left in the joint MoCo training step ("all other kernels").
torch.profiler over one step of tools/time_moco_step.py's shapes; prints the operators / kernels that are not mf:: launches.
MF_FAST=1: the trainer's fast path (loss from the fused partials)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MF_ONLY"] = "none"
here = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(here, "time_moco_step.py")).read().split('if os.environ.get("MF_ONLY") == "step":')[0]
exec(compile(src, "time_moco_step_head", "exec"))
from torch.profiler import profile, ProfilerActivity
step = fwd_bwd_fast if os.environ.get("MF_FAST") == "1" else fwd_bwd
step(); step(); torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=bool(os.environ.get("MF_STACK"))) as prof:
    step(); torch.cuda.synchronize()
ev = prof.key_averages(group_by_stack_n=6 if os.environ.get("MF_STACK") else 0)
kern = [(e.key, e.count, e.device_time_total) for e in ev if e.device_time_total > 0 and e.cpu_time_total == 0]
ops = [(e.key, e.count, e.device_time_total, getattr(e, "stack", None)) for e in ev
       if e.device_time_total > 0 and e.cpu_time_total > 0 and e.key.startswith("aten::")]
print("device kernels that are not mf:: launches")
tot = 0.0
for k, c, t in sorted(kern, key=lambda r: -r[2]):
    if "mf::" in k:
        continue
    tot += t
    print(f"  {k[:90]:90s} x{c:3d} {t:8.1f} us")
print(f"  total {tot:.1f} us / step")
print("aten operators with device time")
for k, c, t, st in sorted(ops, key=lambda r: -r[2])[:40]:
    print(f"  {k:28s} x{c:3d} {t:8.1f} us")
    if st:
        for s in st[:6]:
            if "moco_flow_amd" in s or "tools/" in s:
                print("       ", s)
