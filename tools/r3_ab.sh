#!/bin/bash
# same-box A/B of the forward bench: bash tools/r3_ab.sh "VAR=0" ["VAR2=0" ...]  (each variant vs the default, interleaved twice)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
run() { env $1 timeout 600 python bench.py --no-cpu-baseline --no-pmc --no-train-step --no-secondary --steps 60 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); r=j['roofline']
print('%-28s %7.1f img/s %6.3f ms (%s)  contraction %6.3f ms frac %.4f direct %.3f (%.3f) wino %.3f launches %d' % ('$1', j['value'], j['ms_per_step'], j['launch'][:5], r['kernel_ms_per_step'], r['frac'], r['families']['direct']['ms_per_step'], r['families']['direct']['frac_of_peak'], r['families']['winograd']['ms_per_step'], r['launches_per_step']))"; }
for rep in 1 2; do
  run "DEFAULT=1"
  for v in "$@"; do run "$v"; done
done
