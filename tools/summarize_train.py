#!/usr/bin/env python3
"""Summarise a tools/profile_train.sh output directory: per-kernel time per training step and, for the
mf:: kernels, MFMA utilisation and HBM traffic per dispatch (FETCH_SIZE doubled per the gfx950 note of
MI355X_MICROARCH.md)."""
import csv, glob, os, re, sys
from collections import defaultdict

out = sys.argv[1]
STEPS = 6          # fallback; normally derived below from the backward launches (two NeRF passes per step)


def rows(pattern):
    for f in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r


short = lambda n: re.sub(r"\(.*", "", n.replace("void ", ""))[:60]
print(open(os.path.join(out, "trace.log")).read().strip().splitlines()[-1] if os.path.exists(os.path.join(out, "trace.log")) else "")
dur = defaultdict(list)
for r in rows("trace/**/*kernel_trace.csv"):
    dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in dur.values())
nb = sum(len(v) for k, v in dur.items() if "nerf_backward_kernel" in k)
if nb:
    STEPS = nb / 2
print(f"\n== kernel time per training step (rocprofv3 --kernel-trace; {STEPS} steps) ==  total {tot/STEPS/1e6:.2f} ms/step")
print(f"{'kernel':62s} {'calls/step':>10s} {'avg us':>9s} {'ms/step':>8s} {'%':>6s}")
other = 0.0
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    if k.startswith("mf::") or k.startswith("void mf::") or sum(v) / tot > 0.01:
        print(f"{short(k):62s} {len(v)/STEPS:10.1f} {sum(v)/len(v)/1e3:9.1f} {sum(v)/STEPS/1e6:8.2f} {100*sum(v)/tot:6.1f}")
    else:
        other += sum(v)
print(f"{'(all other kernels: torch elementwise / copies / reductions)':62s} {'':10s} {'':9s} {other/STEPS/1e6:8.2f} {100*other/tot:6.1f}")

pmc = defaultdict(lambda: defaultdict(list))
for sub in ("pmc_mfma", "pmc_fetch", "pmc_write"):
    for r in rows(f"{sub}/**/*counter_collection.csv"):
        if "mf::" in r["Kernel_Name"]:
            pmc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("\n== PMC per dispatch (mean), mf:: kernels ==")
for k, c in sorted(pmc.items(), key=lambda kv: -sum(dur.get(kv[0], [0]))):
    m = {n: sum(v) / len(v) for n, v in c.items()}
    line = f"{short(k):48s}"
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("GRBM_GUI_ACTIVE"):
        cyc = m["GRBM_GUI_ACTIVE"] / 8
        line += f" MFMA busy {m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc) * 100:5.1f} % of SIMD-cycles, {m['SQ_INSTS_VALU_MFMA_F32']:.3g} MFMA,"
    if "FETCH_SIZE" in m:
        line += f" HBM read {2 * m['FETCH_SIZE'] * 1024 / 1e9:.3f} GB (x2-corrected),"
    if "WRITE_SIZE" in m:
        line += f" write {m['WRITE_SIZE'] * 1024 / 1e9:.3f} GB"
    print(line)

# HBM bytes per training step: every kernel's mean FETCH_SIZE (x2: the gfx950 correction of MI355X_MICROARCH.md) +
# WRITE_SIZE per dispatch x its dispatches per step, all kernels of the trace -- and the time-weighted average rate.
fetch, write = defaultdict(list), defaultdict(list)
for r in rows("pmc_fetch/**/*counter_collection.csv"):
    if r["Counter_Name"] == "FETCH_SIZE":
        fetch[r["Kernel_Name"]].append(float(r["Counter_Value"]))
for r in rows("pmc_write/**/*counter_collection.csv"):
    if r["Counter_Name"] == "WRITE_SIZE":
        write[r["Kernel_Name"]].append(float(r["Counter_Value"]))
if fetch and write:
    rd = sum(2 * sum(v) * 1024 for v in fetch.values()) / STEPS
    wr = sum(sum(v) * 1024 for v in write.values()) / STEPS
    ms = tot / STEPS / 1e6
    print(f"\n== HBM traffic per step: read {rd / 1e9:.1f} GB (FETCH_SIZE x2) + write {wr / 1e9:.1f} GB = {(rd + wr) / 1e9:.1f} GB over {ms:.2f} ms of kernels "
          f"= {(rd + wr) / 1e9 / ms:.2f} TB/s average")
    import json
    with open(os.path.join(out, "traffic_entry.json"), "w") as fh:
        json.dump({"hbm_gb_per_step": (rd + wr) / 1e9, "read_gb": rd / 1e9, "write_gb": wr / 1e9, "kernel_ms_per_step": ms}, fh)
    if "--merge" in sys.argv:
        key, tag = sys.argv[sys.argv.index("--merge") + 1], sys.argv[sys.argv.index("--merge") + 2]
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        path = os.path.join(root, "profiles", "traffic.json")
        try:
            allt = json.load(open(path))
        except (OSError, ValueError):
            allt = {}
        allt[key] = {"hbm_gb_per_step": (rd + wr) / 1e9, "read_gb": rd / 1e9, "write_gb": wr / 1e9, "kernel_ms_per_step": ms,
                     "profile": f"profiles/{tag}_summary.txt"}
        json.dump(allt, open(path, "w"), indent=1)

