#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2g
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_contractions.py tests/test_gpu_model.py tests/test_gpu_backward.py -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -6 $O/tests.log
timeout 900 python bench.py --no-cpu-baseline --no-pmc --dump-launches $O/launches.txt > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json,os
j=json.loads(open(os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/r2g/bench.json').read())
r=j['roofline']
print(j['value'], j['ms_per_step'], j['launch'], j['launch_trial'], 'roofline', r['achieved'], r['frac'], r['kernel_ms_per_step'], r['families'], 'train', j['train_step']['ms_per_step'], j['train_step']['launch_trial'])
PY
tail -3 $O/bench.err
