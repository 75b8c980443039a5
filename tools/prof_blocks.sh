#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for nb in 1 2; do
  export MF_BF16_BLOCKS=$nb
  OUT=$REPO/gpurun_out/prof_blocks$nb
  mkdir -p $OUT
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python3 $REPO/bench.py --config C3 --steps 50 --warmup 10 --no-cpu-baseline --no-train-leg --no-extra-legs > $OUT/log 2>&1
  timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/pmc_mix -- python3 $REPO/bench.py --config C3 --steps 50 --warmup 10 --no-cpu-baseline --no-train-leg --no-extra-legs >> $OUT/log 2>&1
  python3 - <<PY
import csv,glob,collections
for sub in ("pmc_mfma","pmc_mix"):
    acc=collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv"%sub, recursive=True):
        for r in csv.DictReader(open(f)):
            if "render_kernel_bf16" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    d=[]
    for f in glob.glob("$OUT/%s/**/*kernel_trace.csv"%sub, recursive=True):
        for r in csv.DictReader(open(f)):
            if "render_kernel_bf16" in r["Kernel_Name"]:
                d.append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
    print("blocks=$nb", sub, "avg_us %.1f n=%d"%(sum(d)/len(d)/1e3,len(d)), {k: "%.4g"%(sum(v)/len(v)) for k,v in acc.items()})
PY
  rm -rf $OUT/pmc_*
done
