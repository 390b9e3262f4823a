#!/bin/bash
# round-2 first GPU call: new parity tests, whole GPU suite, bench (new defaults), power evidence
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2a
mkdir -p $O
cd $R
nproc > $O/host.txt; rocm-smi --showclocks --showpower > $O/smi.txt 2>&1; ls /sys/class/drm/card*/device/hwmon/hwmon*/ >> $O/smi.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q -k "config4 or train_forward or roi_layers or scatter or oracle_same_rng or rccl" > $O/new_tests.log 2>&1; echo "new tests rc=$?" >> $O/new_tests.log
tail -5 $O/new_tests.log
timeout 2400 python -m pytest tests -m gpu -q > $O/all_tests.log 2>&1; echo "all tests rc=$?" >> $O/all_tests.log
tail -8 $O/all_tests.log
timeout 900 python bench.py --dump-launches $O/launches.txt > $O/bench.json 2> $O/bench.err; cut -c1-600 $O/bench.json; tail -3 $O/bench.err
timeout 600 python tools/gemm_power.py $O/gemm_power.md > $O/gemm_power.log 2>&1; tail -12 $O/gemm_power.log
