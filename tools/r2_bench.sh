#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2d
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_graphs.py tests/test_gpu_model.py -x -q -k "graph or train_forward" > $O/graphs.log 2>&1; echo "rc=$?" >> $O/graphs.log; tail -5 $O/graphs.log
timeout 900 python bench.py --dump-launches $O/launches.txt > $O/bench.json 2> $O/bench.err; cut -c1-1000 $O/bench.json; tail -5 $O/bench.err
timeout 600 python bench.py --mode step --batch 1 --no-cpu-baseline > $O/bench_step_b1.json 2> $O/bench_step_b1.err; cut -c1-900 $O/bench_step_b1.json; tail -3 $O/bench_step_b1.err
timeout 600 python bench.py --mode step --no-cpu-baseline > $O/bench_step.json 2>/dev/null; cut -c1-900 $O/bench_step.json
