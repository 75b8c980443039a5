#!/usr/bin/env python3
"""profiles/traffic.json <- gpurun_out/<round>/<round>_<CFG>_traffic_entry.json (what tools/profile_all.sh leaves) + copy the
summaries / kernel stats into profiles/ under lower-case names.  usage: tools/merge_traffic.py r06 C2 C3 ... [train_joint ...]"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, names = sys.argv[1], sys.argv[2:]
path = os.path.join(ROOT, "profiles", "traffic.json")
allt = json.load(open(path))
for n in names:
    src = os.path.join(ROOT, "gpurun_out", rnd)
    tag = f"{rnd}_{n.lower()}"
    entry = json.load(open(os.path.join(src, f"{rnd}_{n}_traffic_entry.json")))
    entry["profile"] = f"profiles/{tag}_summary.txt"
    if n.startswith("train_"):
        allt[n] = entry
    else:
        entry["source"] = (f"profiles/{tag}_summary.txt (rocprofv3 --kernel-trace --stats + --pmc passes of `bench.py --config {n}`; FETCH x2 per "
                           f"MI355X_MICROARCH.md; median over the dispatches)")
        allt[n] = entry
    shutil.copy(os.path.join(src, f"{rnd}_{n}_summary.txt"), os.path.join(ROOT, "profiles", f"{tag}_summary.txt"))
    ks = os.path.join(src, f"{rnd}_{n}_kernel_stats.csv")
    if os.path.exists(ks):
        shutil.copy(ks, os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
    print("merged", n, "->", tag)
json.dump(allt, open(path, "w"), indent=1)
