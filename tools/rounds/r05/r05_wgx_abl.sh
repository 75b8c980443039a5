# timing ablations of wg_segment_x3 (apply tools/variants/wgrad_x3_ablation_r05.patch to csrc/mf_wgrad.hip first, then
# tools/ab_lib.sh build mf_wgrad.hip "a2=-DMF_WGX_ABL=2" ...; build/ab/lib_<name>.so = -DMF_WGX_ABL=<bits>: 1 no MFMA, 2 no re-loads, 4 no fragment reads,
# 8 no conversion arithmetic, 16 no fragment writes, 32 no barriers); results of the ablated libraries are garbage
for rep in 1 2; do for n in "$@"; do L=$GRAFT_REPO_ROOT/build/ab/lib_$n.so; [ $n = default ] && L=""; echo "== $n"; MOCOFLOW_HIP_LIB=$L MF_WGRAD=bf16x3 MF_WGRAD_SETS="A x9,F x3" timeout 200 python tools/bench_wgrad.py 2>&1 | grep " ms "; done; done
