#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2c
mkdir -p $O
cd $R
for m in 8 5 4; do
  timeout 300 python tools/graph_debug.py $m > $O/dbg$m.log 2>&1; echo "mode $m rc=$?"; grep -E "OK|Error|error|File ./tmp/code" $O/dbg$m.log | head -12
done
timeout 1200 python -m pytest tests/test_gpu_graphs.py tests/test_gpu_backward.py tests/test_gpu_trainer_dist.py -x -q > $O/graphs.log 2>&1; echo "rc=$?" >> $O/graphs.log; tail -15 $O/graphs.log
