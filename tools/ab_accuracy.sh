for l in nohw hw; do for c in C2b C3 C5; do
MOCOFLOW_HIP_LIB=build/ab/lib_$l.so python3 bench.py --config $c --steps 20 --warmup 3 --no-train-leg --no-extra-legs 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; e=d.get('error_vs_cpu',{})
print('$l $c kernel_ms %.4f frac %.3f' % (r['kernel_ms'], r['frac']), 'psnr', e.get('psnr_equiv_db'), 'l2', {k: round(v,5) for k,v in e.get('l2_rel',{}).items()})"
done; done
