#!/bin/bash
# K fresh processes of the eager forward under rocprofv3 --kernel-trace on one box: per process the step span and which
# hardware queue each role landed on (tools/queue_map.py) -- do the fast and the slow population of
# profiles/r5_step_time_populations.txt differ in their queue assignment?   usage: bash tools/populations_probe.sh [K] [steps]
R=${GRAFT_REPO_ROOT:-/root/repo}
K=${1:-5}; STEPS=${2:-30}
O=$R/gpurun_out/populations; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for i in $(seq 1 $K); do
  rm -rf /tmp/rp_pop
  rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_pop -o pop -- python $R/bench.py --launch eager --steps $STEPS --warmup 5 \
    --no-cpu-baseline --no-roofline --no-train-step --no-secondary --no-pmc > $O/run_$i.log 2>&1
  f=$(find /tmp/rp_pop -name "*kernel_trace.csv" | head -1)
  echo "process $i: bench line $(grep '^{' $O/run_$i.log | tail -1 | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('%.1f img/s %.3f ms (under the profiler)' % (j['value'], j['ms_per_step']))" 2>/dev/null)"
  python $R/tools/queue_map.py $f
done | tee $O/populations.txt
