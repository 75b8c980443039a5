#!/usr/bin/env python3
"""profiles/kernel_resources.json: VGPRs / AGPRs / SGPRs / occupancy / scratch / LDS of every kernel of the library AS THE
COMPILER REPORTS THEM (hipcc -Rpass-analysis=kernel-resource-usage on each translation unit with the unit's own flags of the
shipped build, csrc/Makefile), keyed by the demangled kernel name rocprofv3 prints.  tools/summarize_prof.py reads it for
the register line of a profile summary (rocprofv3's VGPR_Count / Accum_VGPR_Count columns are allocation granules of the
dispatch packet, not these numbers: VERDICT r4).  CPU only, ~2 min.  usage: tools/kernel_resources.py [unit ...]"""
import json
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "moco_flow_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
FILT = "/usr/bin/c++filt"


def compile_unit(unit):
    flags = subprocess.run(["make", "-s", "unitflags", f"UNIT={unit}"], cwd=CSRC, capture_output=True, text=True, check=True).stdout.split()
    out = f"/tmp/mf_res_{os.getpid()}_{unit}.s"
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--cuda-device-only",
                        "-Rpass-analysis=kernel-resource-usage", "-S", "-o", out] + flags + [unit + ".hip"],
                       cwd=CSRC, capture_output=True, text=True, timeout=1800)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-2000:])
    os.remove(out)
    usage, name = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            usage[name] = {"unit": unit}
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[(?:bytes/lane|waves/SIMD|bytes/block)\])?: (\d+)", line)
        if m and name:
            usage[name][m.group(1).strip()] = int(m.group(2))
    return usage


def main():
    units = sys.argv[1:] or subprocess.run(["make", "-s", "units"], cwd=CSRC, capture_output=True, text=True, check=True).stdout.split()
    with ThreadPoolExecutor(4) as ex:
        parts = list(ex.map(compile_unit, units))
    usage = {}
    for p in parts:
        usage.update(p)
    names = list(usage)
    dem = subprocess.run([FILT], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    path = os.path.join(ROOT, "profiles", "kernel_resources.json")
    try:
        out = json.load(open(path)) if sys.argv[1:] else {}
    except (OSError, ValueError):
        out = {}
    for n, d in zip(names, dem):
        out[d] = dict(usage[n], mangled=n)
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    for d in sorted(out):
        u = out[d]
        if u.get("VGPRs", 0) >= 96:
            print(f"{d[:100]:100s} VGPR {u.get('VGPRs')} AGPR {u.get('AGPRs')} occ {u.get('Occupancy')} scratch {u.get('ScratchSize')} LDS {u.get('LDS Size')}")


if __name__ == "__main__":
    main()
