#!/bin/bash
# Race / hazard screen of the fused kernels (run ON THE GPU BOX): tools/stress_determinism.py <precision> 20 -- every case repeated 20
# times, bitwise -- with the shipped library and with the whole library built -DMF_DBG_JITTER (random per-wave stalls in front of every
# panel barrier: build/ab/lib_jit.so from `MF_VARIANT_FLAGS=-DMF_DBG_JITTER MF_VARIANT_NAME=jit tools/build_timeline.sh`), both kernel
# families of the fast mode (MF_BF16_BLOCKS=1|2).
cd ${GRAFT_REPO_ROOT:-.}
for lib in default jit; do
  L=""; [ $lib = jit ] && L=build/ab/lib_jit.so
  for prec in bf16 bf16x3 f32; do
    echo "== $lib $prec"; MOCOFLOW_HIP_LIB=$L timeout 400 python3 tools/stress_determinism.py $prec 20 2>&1 | grep -E "deterministic|DIFFERS"
  done
  echo "== $lib bf16, two-block family (MF_BF16_BLOCKS=2)"; MF_BF16_BLOCKS=2 MOCOFLOW_HIP_LIB=$L timeout 400 python3 tools/stress_determinism.py bf16 20 2>&1 | grep -E "deterministic|DIFFERS"
done
