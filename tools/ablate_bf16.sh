for f in 0 1 2 3 19; do
  echo -n "bf16 MF_DEBUG_FLAGS=$f: "
  MF_DEBUG_FLAGS=$f python bench.py --no-cpu-baseline --no-train-leg --precision bf16 --steps 30 --warmup 3 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ms', round(d['ms_per_step'],4), 'kernel', round(d['roofline']['kernel_ms'],4))"
done
