#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2b
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_graphs.py -x -q > $O/graphs.log 2>&1; echo "rc=$?" >> $O/graphs.log; tail -30 $O/graphs.log
timeout 1200 python -m pytest tests -m gpu -x -q -k "proposal_target_hip or sibling_backward or test_gpu_model or trainer" > $O/re.log 2>&1; echo "rc=$?" >> $O/re.log; tail -8 $O/re.log
