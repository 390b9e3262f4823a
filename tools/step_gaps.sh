#!/bin/bash
# GPU-idle gaps of the eager training iteration (or the forward: MODE=train) from a rocprofv3 kernel trace.
# usage: step_gaps.sh [min_gap_us]
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
MODE=${MODE:-step}
cd /tmp; rm -rf /tmp/sg
rocprofv3 --kernel-trace --output-format csv -d /tmp/sg -o t -- python $R/bench.py --launch eager --mode $MODE --steps 6 --warmup 3 \
  --no-cpu-baseline --no-roofline --no-train-step --no-secondary --no-pmc > /tmp/sg.log 2>&1
T=$(find /tmp/sg -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_gaps.py $T ${MARKER:-sgd_momentum} | head -60
