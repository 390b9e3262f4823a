#!/bin/bash
# usage: exposed_intervals.sh <mode: step|train> <launch> [marker] [min_us]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
MODE=${1:-step}; LAUNCH=${2:-program}; MARK=${3:-sgd_momentum}; MINUS=${4:-25}
cd /tmp; rm -rf /tmp/ei
rocprofv3 --kernel-trace --output-format csv -d /tmp/ei -o t -- python $R/bench.py --launch $LAUNCH --mode $MODE --steps 6 --warmup 3 \
  --no-cpu-baseline --no-roofline --no-train-step --no-secondary --no-pmc > /tmp/ei.log 2>&1
T=$(find /tmp/ei -name "*kernel_trace.csv" | head -1)
python $R/tools/exposed_intervals.py $T $MARK $MINUS
