for d in randn zero relu randn zero relu; do echo "== $d"; MF_WGRAD_DATA=$d MF_WGRAD=bf16x3 MF_WGRAD_SETS="A x9,F x3,NoF" timeout 200 python tools/bench_wgrad.py 2>&1 | grep " ms "; done
