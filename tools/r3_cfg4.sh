#!/bin/bash
# rocprofv3 evidence for BASELINE configs[4]'s per-GPU shape (800x1333 queries, shot 10, 2 episodes per GPU):
# kernel stats (single stream: every kernel alone on the chip), HBM traffic (FETCH_SIZE / WRITE_SIZE passes) and
# MFMA-busy counters -> gpurun_out/r3_cfg4/. Run on the GPU box: bash tools/r3_cfg4.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3_cfg4
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CFG="--height 800 --width 1333 --shot 10 --batch 2"
COMMON="--launch eager --no-cpu-baseline --no-roofline --no-train-step --no-secondary --no-pmc"
for mode in single_stream default; do
  extra=""; [ $mode = single_stream ] && extra="--single-stream"
  rm -rf /tmp/rp_$mode
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$mode -o bench -- python $R/bench.py $CFG $COMMON --steps 10 --warmup 3 $extra > $O/$mode.log 2>&1
  cp $(find /tmp/rp_$mode -name "*kernel_stats.csv" | head -1) $O/${mode}_kernel_stats.csv
  tail -1 $O/$mode.log | cut -c1-200
done
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o pmc -- python $R/bench.py $CFG $COMMON --steps 3 --warmup 1 --single-stream > $O/pmc_$c.log 2>&1
  cp $(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1) /tmp/pmc_$c.csv
done
python $R/tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE.csv /tmp/pmc_WRITE_SIZE.csv 4 $O/pmc_traffic.json > $O/pmc_traffic.txt
rm -rf /tmp/pmc_mfma
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_mfma -o pmc -- python $R/bench.py $CFG $COMMON --steps 3 --warmup 1 --single-stream > $O/pmc_mfma.log 2>&1
cp $(find /tmp/pmc_mfma -name "*counter_collection.csv" | head -1) $O/pmc_mfma_counter_collection.csv
cd $R && python bench.py $CFG --no-cpu-baseline --no-pmc --dump-launches $O/launches.txt > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
