cd $GRAFT_REPO_ROOT
# correctness of each variant first (the oracle-of-the-arithmetic tests: fast + x3), then timing
for L in default f442 f422 f242; do
  LIB=$PWD/build/ab/lib_$L.so; [ $L = default ] && LIB=""
  echo "=== $L"; MOCOFLOW_HIP_LIB=$LIB timeout 900 python -m pytest tests/test_gpu_bf16_oracle.py -m gpu -x -q 2>&1 | tail -3
done
bash tools/ab_run.sh "C3 C3g C5 C3x" base default f442 f422 f242 2>&1 | tee gpurun_out/r05_ab_noftpp.txt
