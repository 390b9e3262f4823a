#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2h
mkdir -p $O
cd $R
timeout 3000 python -m pytest tests -m gpu -q > $O/all_tests.log 2>&1; echo "all tests rc=$?" >> $O/all_tests.log
tail -8 $O/all_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-700 $O/bench.json; tail -3 $O/bench.err
