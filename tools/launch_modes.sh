run() { python bench.py --no-cpu-baseline --no-pmc --no-train-step --no-secondary --no-roofline --steps 60 "$@" 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read())
print('%-40s %7.1f img/s %6.3f ms  host %s  (%s)' % ('$*', j['value'], j['ms_per_step'], j['host_enqueue_ms_per_step'], j['launch'][:24]))"; }
for rep in 1 2; do
run --launch eager
run --launch graph
run --launch eager --device-rng
run --launch graph --device-rng
done
