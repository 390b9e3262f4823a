for v in 1 0; do DANA_WGRAD_PIPELINED=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-pmc --no-secondary --launch eager 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); t=j['train_step']; r=t['roofline']
print('pipelined=$v train', t['ms_per_step'], 'kernel-alone sum', r['kernel_ms_per_step'], 'frac', r['frac'], {k:(v['ms_per_step'],v['frac_of_peak']) for k,v in r['families'].items()})"; done
