cd $GRAFT_REPO_ROOT
b() { python bench.py --no-cpu-baseline --no-roofline --no-train-step "$@" 2>&1 | tail -1 | python -c "
import sys, json
l=sys.stdin.read().strip()
try:
    r=json.loads(l); print('%-60s %8.1f img/s %7.3f ms' % (' '.join(sys.argv[1:]), r['value'], r['ms_per_step']))
except Exception as e:
    print('FAILED', ' '.join(sys.argv[1:]), l[-300:])
" "$@"; }
b --mode eval
b --mode eval --batch 1
b --mode infer --batch 1
b --ba
b --mode step
b --mode step --batch 1
b --mode step --batch 2
b --model frcnn
b --model meta
b --model fsod
b --model fgn
b --support-size 224
b --mode step --support-size 224
b --mode step --batch 2 --height 800 --width 1333 --shot 10
