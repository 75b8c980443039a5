#!/bin/bash
# counters of the x3 training forward (render_kernel_bf16<*, true, true>) beside the gradient-free pass, tools/time_dump_pass.py
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE" \
           "SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE" \
           "SQC_ICACHE_MISSES SQC_ICACHE_HITS SQC_ICACHE_REQ SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/dp_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/dp_$i -- python3 $REPO/tools/time_dump_pass.py 1024 > /tmp/dp_$i.log 2>&1
done
python3 - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for i in (1, 2, 3, 4):
    for f in glob.glob(f"/tmp/dp_{i}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "render_kernel" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:64] + " grid" + r.get("Grid_Size", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print("==", k)
    for c, v in sorted(acc[k].items()):
        print(f"   {c:32s} {sum(v)/len(v):14.5g}  (n={len(v)})")
PY
