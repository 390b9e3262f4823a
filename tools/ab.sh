#!/bin/bash
# same-box A/B of the forward bench: bash tools/ab.sh "VAR=0" ["VAR2=0 VAR3=1" ...]  -- each variant (a set of environment
# switches; DANA_LIB_PATH=<other libdana_hip.so> compares two builds) against the default, interleaved, REPS times
# (replaces the per-round r2_ab.sh / r3_ab.sh / r3_fuse_ab.sh scripts). MODE=step benches the training iteration.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
REPS=${REPS:-2}; STEPS=${STEPS:-60}; MODE=${MODE:-train}
run() { env $1 timeout ${TMO:-900} python bench.py --mode $MODE --no-cpu-baseline --no-pmc --no-train-step --no-secondary --steps $STEPS $EXTRA 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j.get('roofline') or {}
f=r.get('families',{})
print('%-34s %7.1f img/s %6.3f ms median %6.3f (%s)  contraction %s ms frac %s direct %s wino %s launches %s' % ('$1', j['value'], j['ms_per_step'], j['ms_per_step_median'] or 0, j['launch'][:5], r.get('kernel_ms_per_step'), r.get('frac'), (f.get('direct') or {}).get('ms_per_step'), (f.get('winograd') or {}).get('ms_per_step'), r.get('launches_per_step')))"; }
for rep in $(seq 1 $REPS); do
  run "DEFAULT=1"
  for v in "$@"; do run "$v"; done
done
