#!/usr/bin/env python3
"""Issue-slot budget per MFMA gap of a kernel, from its ISA (CPU only).

usage: tools/issue_budget.py <unit> <kernel-name-fragment> [...]      e.g.  mf_render_bf16 'render_kernel_bf16ILb1ELb1ELb0E'

Compiles csrc/<unit>.hip to assembly with the unit's own flags (csrc/Makefile `unitflags`), takes the kernel whose mangled name
contains the fragment, and walks its instruction stream: a GAP is what stands between two consecutive matrix instructions.  Per
gap the instructions are counted by class -- other VALU (converts, max, FMAs ...), v_accvgpr moves, LDS reads, other LDS, LDS-DMA
/ VMEM, SALU, s_waitcnt, s_nop (with its wait states), branches / barriers -- and reported

  * over ALL gaps, and
  * over the DENSE gaps (<= 12 instructions: the tile loops; the long gaps are the VALU phases between the networks),

as mean per gap and as a histogram of the gap's total issue count, next to the guide's budget: a 32x32x16 MFMA occupies the
matrix pipe for 8 passes = 32 cycles, a wave64 VALU instruction issues in 4, so a wave ALONE on its SIMD (the bf16x3 kernels)
hides at most ~7 four-cycle issues behind each MFMA before the pipe runs dry (MI355X_MICROARCH.md; ~5 once waits and the
MFMA's own issue are counted), two waves per SIMD (the fast mode) twice that between them."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "moco_flow_amd", "csrc")


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_accvgpr"):
        return "accvgpr"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "lds_read"
    if op.startswith("ds_"):
        return "lds_other"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("scratch_") or op.startswith("flat_"):
        return "vmem"
    if op == "s_waitcnt":
        return "waitcnt"
    if op == "s_nop":
        return "nop"
    if op in ("s_barrier",) or op.startswith("s_cbranch") or op == "s_branch":
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    unit, frags = sys.argv[1], sys.argv[2:]
    flags = subprocess.run(["make", "-s", "unitflags", f"UNIT={unit}"], cwd=CSRC, capture_output=True, text=True, check=True).stdout.split()
    out = f"/tmp/issue_budget_{unit}.s"
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(os.path.join(CSRC, f)) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))):
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--cuda-device-only",
                        "-S", "-o", out] + flags + [unit + ".hip"], cwd=CSRC, check=True)
    text = open(out).read()
    for frag in frags:
        m = re.search(r"^(_Z\w*%s\w*):[^\n]*\n" % re.escape(frag), text, re.M)
        if not m:
            print("kernel not found:", frag)
            continue
        name = m.group(1)
        body = text[m.end():text.index("s_endpgm", m.end())]
        gaps, cur, nopw = [], collections.Counter(), 0
        seen_mfma = False
        for line in body.splitlines():
            line = line.split(";")[0].strip()
            if not line or line.endswith(":") or line.startswith("."):
                continue
            op = line.split()[0]
            c = classify(op)
            if c == "mfma":
                if seen_mfma:
                    gaps.append(cur)
                cur, seen_mfma = collections.Counter(), True
                continue
            if seen_mfma:
                cur[c] += 1
                if c == "nop":
                    cur["nop_states"] += 1 + int(line.split()[1])
        classes = ["valu", "accvgpr", "lds_read", "lds_other", "vmem", "salu", "waitcnt", "nop", "nop_states", "branch", "other"]
        tot = lambda g: sum(v for k, v in g.items() if k != "nop_states")
        dense = [g for g in gaps if tot(g) <= 12]
        print(f"== {name}\n   {len(gaps) + 1} matrix instructions, {len(gaps)} gaps, {len(dense)} dense (<= 12 issues)")
        for label, gs in (("all gaps", gaps), ("dense gaps", dense)):
            n = max(len(gs), 1)
            print(f"   {label:10s} mean issues per gap: total {sum(tot(g) for g in gs) / n:5.2f} | " +
                  "  ".join(f"{c} {sum(g[c] for g in gs) / n:.2f}" for c in classes))
        hist = collections.Counter(min(tot(g), 13) for g in gaps)
        print("   histogram of issues per gap (13 = more): " + "  ".join(f"{k}:{hist[k]}" for k in sorted(hist)))
        # issue time of a dense gap: 4 cycles per VALU / accvgpr / LDS / VMEM issue, 1 per SALU-class, the nop's wait states
        cyc = [4 * (g["valu"] + g["accvgpr"] + g["lds_read"] + g["lds_other"] + g["vmem"]) + g["salu"] + g["waitcnt"] + g["branch"] + g["nop_states"] for g in dense]
        over = sum(1 for c in cyc if c > 28)
        print(f"   dense gaps: estimated issue cycles per gap mean {sum(cyc) / max(len(cyc), 1):.1f} of the 32 a 32x32x16 MFMA covers (28 beside its own issue); "
              f"{over} of {len(cyc)} gaps ({100.0 * over / max(len(cyc), 1):.1f} %) exceed 28")


if __name__ == "__main__":
    main()
