#!/usr/bin/env python3
"""Summarise a tools/profile_gpu.sh output directory: per-kernel stats + PMC counters averaged
per dispatch of the dominant kernel. Applies the gfx950 FETCH_SIZE x2 correction of
MI355X_MICROARCH.md (HBM section) and reports both raw and corrected values."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def rows(pattern):
    for f in glob.glob(os.path.join(out, pattern), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                yield r


print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
for f in glob.glob(os.path.join(out, "trace/**/*kernel_stats.csv"), recursive=True):
    with open(f) as fh:
        for i, line in enumerate(fh):
            if i < 12:
                print(line.rstrip())
dur = defaultdict(list)
by_grid = defaultdict(lambda: defaultdict(list))        # kernel -> grid size -> durations
for r in rows("trace/**/*kernel_trace.csv"):
    dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    by_grid[r["Kernel_Name"]][(r.get("Grid_Size_X", r.get("Grid_Size")), r.get("LDS_Block_Size"))].append(dur[r["Kernel_Name"]][-1])


def two_clusters(d):
    """A kernel whose dispatches fall into two duration classes (coarse and fine pass of one call: same grid -- one workgroup per
    CU -- and dynamic LDS the trace does not record): 2-means on log-duration; None when the spread is within a factor 1.6."""
    import math
    if len(d) < 4 or max(d) < 1.6 * min(d):
        return None
    lo, hi = math.log(min(d)), math.log(max(d))
    for _ in range(20):
        a = [x for x in d if abs(math.log(x) - lo) <= abs(math.log(x) - hi)]
        b = [x for x in d if abs(math.log(x) - lo) > abs(math.log(x) - hi)]
        if not a or not b:
            return None
        lo, hi = sum(math.log(x) for x in a) / len(a), sum(math.log(x) for x in b) / len(b)
    return a, b
dom = None
if dur:
    dom = max(dur, key=lambda k: sum(dur[k]))
    d = dur[dom]
    print(f"\ndominant kernel: {dom[:90]}\n  dispatches {len(d)}  avg {sum(d)/len(d)/1e3:.1f} us  min {min(d)/1e3:.1f}  max {max(d)/1e3:.1f}")
    # the same kernel at DIFFERENT launch shapes is different work (a coarse and a fine pass of one render_rays call, the
    # coarse / fine launches of a training step): one line per (grid, LDS) shape, so that an average never mixes them
    if len(by_grid[dom]) > 1:
        for (grid, lds), dd in sorted(by_grid[dom].items(), key=lambda kv: -sum(kv[1])):
            print(f"    grid {grid} lds {lds}: dispatches {len(dd)}  avg {sum(dd)/len(dd)/1e3:.1f} us  min {min(dd)/1e3:.1f}  max {max(dd)/1e3:.1f}"
                  f"  ({100.0*sum(dd)/sum(d):.0f} % of the kernel's time)")
    else:
        cl = two_clusters(d)
        if cl:
            for label, dd in (("short launches (coarse pass)", cl[0]), ("long launches (fine pass)", cl[1])):
                print(f"    {label}: dispatches {len(dd)}  avg {sum(dd)/len(dd)/1e3:.1f} us  min {min(dd)/1e3:.1f}  max {max(dd)/1e3:.1f}"
                      f"  ({100.0*sum(dd)/sum(d):.0f} % of the kernel's time)")
    r0 = next(r for r in rows("trace/**/*kernel_trace.csv") if r["Kernel_Name"] == dom)
    # registers / occupancy as the COMPILER reports them (profiles/kernel_resources.json, tools/kernel_resources.py); the
    # trace's VGPR_Count / Accum_VGPR_Count columns are the dispatch packet's allocation fields, not these numbers
    res = {}
    try:
        import json as _json
        allres = _json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "kernel_resources.json")))
        norm = lambda s: s.replace("void ", "").replace(" ", "")
        res = next((v for k, v in allres.items() if norm(k) == norm(dom)), {})
    except (OSError, ValueError):
        pass
    if res:
        print(f"  hipcc resource usage: VGPRs {res.get('VGPRs')}  AGPRs {res.get('AGPRs')}  SGPRs {res.get('SGPRs')}  occupancy {res.get('Occupancy')} "
              f"waves/SIMD  scratch {res.get('ScratchSize')} B/lane  LDS {res.get('LDS Size')} B static  (VGPR spills {res.get('VGPRs Spill')})")
    else:
        print("  hipcc resource usage: not found in profiles/kernel_resources.json (run tools/kernel_resources.py)")
    print("  dispatch packet: LDS", r0.get("LDS_Block_Size"), "scratch", r0.get("Scratch_Size"), "grid", r0.get("Grid_Size_X", r0.get("Grid_Size")),
          "wg", r0.get("Workgroup_Size_X", r0.get("Workgroup_Size")))
print("\n== PMC (per dispatch of the dominant kernel, mean) ==")
for sub in ("pmc_mfma", "pmc_fetch", "pmc_write", "pmc_wait"):
    acc = defaultdict(list)
    for r in rows(f"{sub}/**/*counter_collection.csv"):
        if dom and r["Kernel_Name"] != dom:
            continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{sub:10s} {k:28s} mean {sum(v)/len(v):.6g}  (n={len(v)})")
        if k == "FETCH_SIZE":
            kb = sum(v) / len(v)
            print(f"{'':10s} -> HBM read  raw {kb*1024/1e6:.3f} MB, gfx950-corrected (x2) {2*kb*1024/1e6:.3f} MB per launch")
        if k == "WRITE_SIZE":
            kb = sum(v) / len(v)
            print(f"{'':10s} -> HBM write {kb*1024/1e6:.3f} MB per launch (uncalibrated)")
        if k in ("FETCH_SIZE", "WRITE_SIZE"):
            # per-dispatch distribution: a mean can hide a few dispatches of a different kind (planes requested, a graph replay)
            sv = sorted(v)
            hist = defaultdict(int)
            for x in v:
                hist[int(round(x / 64.0)) * 64] += 1
            top = sorted(hist.items(), key=lambda kv: -kv[1])[:4]
            print(f"{'':10s}    per dispatch (KiB): min {sv[0]:.0f}  median {sv[len(sv)//2]:.0f}  max {sv[-1]:.0f};  most frequent (64-KiB bins): "
                  + ", ".join(f"{b} x{c}" for b, c in top))

# traffic.json entry for bench.py's roofline.traffic (HBM bytes per launch of the dominant kernel: FETCH_SIZE x2 per
# the gfx950 note + WRITE_SIZE, both in KiB units as rocprofv3 reports them; median over the dispatches)
import json
vals = {}
for sub in ("pmc_fetch", "pmc_write"):
    acc = defaultdict(list)
    for r in rows(f"{sub}/**/*counter_collection.csv"):
        if dom and r["Kernel_Name"] != dom:
            continue
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        # MEDIAN per dispatch: the first dispatch of a process can carry two thousand times the bytes of the others (r04_c2:
        # 86 dispatches at 80 KiB of WRITE_SIZE, one at 168 625 KiB -- round 3's "2.07 MB per launch" was that one in the mean)
        vals[k] = sorted(v)[len(v) // 2]
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    entry = {"bytes_per_launch": (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024,
             "fetch_kib_raw": vals["FETCH_SIZE"], "write_kib": vals["WRITE_SIZE"], "kernel": dom}
    # round 5: what bench.py needs for roofline.frac_rocprof_avg / mfma_busy / ghz -- the kernel-trace average of the dominant
    # kernel and, from the pmc_mfma pass, MFMA busy (SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)) and the
    # effective clock of THAT pass (cycles per XCD / its own average duration)
    if dom and dur.get(dom):
        d = dur[dom]
        entry["rocprof_avg_us"] = sum(d) / len(d) / 1e3
        entry["rocprof_min_us"] = min(d) / 1e3
        entry["dispatches"] = len(d)
        if len(by_grid[dom]) > 1:
            entry["by_grid"] = {str(g[0]): {"n": len(dd), "avg_us": sum(dd) / len(dd) / 1e3, "min_us": min(dd) / 1e3}
                                for g, dd in by_grid[dom].items()}
        elif two_clusters(d):
            a, b = two_clusters(d)
            entry["by_pass"] = {"coarse": {"n": len(a), "avg_us": sum(a) / len(a) / 1e3, "min_us": min(a) / 1e3},
                                "fine": {"n": len(b), "avg_us": sum(b) / len(b) / 1e3, "min_us": min(b) / 1e3}}
    acc, pdur = defaultdict(list), []
    for r in rows("pmc_mfma/**/*counter_collection.csv"):
        if r["Kernel_Name"] == dom:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for r in rows("pmc_mfma/**/*kernel_trace.csv"):
        if r["Kernel_Name"] == dom:
            pdur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    if acc.get("GRBM_GUI_ACTIVE") and acc.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        cyc = sum(acc["GRBM_GUI_ACTIVE"]) / len(acc["GRBM_GUI_ACTIVE"]) / 8
        entry["mfma_busy"] = sum(acc["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(acc["SQ_VALU_MFMA_BUSY_CYCLES"]) / (1024 * cyc)
        entry["cycles_per_xcd"] = cyc
        if pdur:
            entry["ghz"] = cyc / (sum(pdur) / len(pdur))
    with open(os.path.join(out, "traffic_entry.json"), "w") as fh:
        json.dump(entry, fh)
    print("\ntraffic entry:", json.dumps(entry))
    # --merge KEY TAG: profiles/traffic.json[KEY] <- this entry (+ its source), profiles/<TAG>_summary.txt is this output's name
    if "--merge" in sys.argv:
        key, tag = sys.argv[sys.argv.index("--merge") + 1], sys.argv[sys.argv.index("--merge") + 2]
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        path = os.path.join(root, "profiles", "traffic.json")
        try:
            allt = json.load(open(path))
        except (OSError, ValueError):
            allt = {}
        entry["profile"] = f"profiles/{tag}_summary.txt"
        entry["source"] = (f"profiles/{tag}_summary.txt (rocprofv3 --kernel-trace --stats + --pmc passes of `bench.py --config {key}`; FETCH x2 per "
                           f"MI355X_MICROARCH.md; median over the dispatches)")
        allt[key] = entry
        json.dump(allt, open(path, "w"), indent=1)
