#!/bin/bash
# usage: exposed_time.sh <mode: step|train> <launch: eager|program|graph> [marker]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
MODE=${1:-step}; LAUNCH=${2:-program}; MARK=${3:-sgd_momentum}
cd /tmp; rm -rf /tmp/et
rocprofv3 --kernel-trace --output-format csv -d /tmp/et -o t -- python $R/bench.py --launch $LAUNCH --mode $MODE --steps 6 --warmup 3 \
  --no-cpu-baseline --no-roofline --no-train-step --no-secondary --no-pmc > /tmp/et.log 2>&1
T=$(find /tmp/et -name "*kernel_trace.csv" | head -1)
python $R/tools/exposed_time.py $T $MARK
