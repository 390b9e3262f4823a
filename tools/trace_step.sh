#!/bin/bash
# one step's kernel timeline of the default (two-stream, eager) forward: rocprofv3 --kernel-trace, then the rows of the
# last complete step (marker = rcnn_loss_b_kernel) -> $O/step_trace.csv   usage: bash tools/trace_step.sh <outdir> [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${1:-r4a}; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_tl -o tl -- python $R/bench.py --launch eager --steps 8 --warmup 3 \
  --no-cpu-baseline --no-roofline --no-train-step --no-secondary --no-pmc "$@" > $O/trace_bench.log 2>&1
f=$(find /tmp/rp_tl -name "*kernel_trace.csv" | head -1)
python $R/tools/step_timeline.py $f $O/step_trace.csv > $O/timeline.txt
tail -5 $O/timeline.txt
