#!/bin/bash
# Run ON THE GPU BOX (through gpurun): instruction mix + wave-state PMC passes of bench.py (no tracing domains).
# usage: tools/profile_mix.sh <tag> [bench args...]     -> gpurun_out/mix_<tag>/summary.txt
set -u
TAG=${1:-mix}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/mix_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 3 --no-cpu-baseline --no-train-leg --no-extra-legs $*"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_BRANCH --output-format csv -d $OUT/p1 -- python3 $REPO/bench.py $ARGS > $OUT/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/p2 -- python3 $REPO/bench.py $ARGS > $OUT/p2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_CVT SQ_INSTS_VALU_FMA_F32 --output-format csv -d $OUT/p3 -- python3 $REPO/bench.py $ARGS > $OUT/p3.log 2>&1
python3 - $OUT > $OUT/summary.txt <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
tot = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "p*/**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        tot[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dom = max(tot, key=lambda k: sum(tot[k].get("SQ_WAVE_CYCLES", tot[k].get("SQ_INSTS_VALU", [0]))))
print("kernel:", dom[:100])
c = {k: sum(v) / len(v) for k, v in tot[dom].items()}
for k in sorted(c):
    print(f"{k:28s} {c[k]:.6g}")
if "GRBM_GUI_ACTIVE" in c:
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    print(f"cycles/XCD {cyc:.4g}; MFMA busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * cyc):.3f} of SIMD-cycles")
if "SQ_INSTS_MFMA" in c:
    print(f"per MFMA: VALU {c['SQ_INSTS_VALU'] / c['SQ_INSTS_MFMA']:.2f} LDS {c['SQ_INSTS_LDS'] / c['SQ_INSTS_MFMA']:.2f} "
          f"SALU {c['SQ_INSTS_SALU'] / c['SQ_INSTS_MFMA']:.2f} VMEM {c['SQ_INSTS_VMEM'] / c['SQ_INSTS_MFMA']:.3f}")
PY
cat $OUT/summary.txt
