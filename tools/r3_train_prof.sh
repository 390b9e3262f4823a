#!/bin/bash
# kernel stats of the training iteration (variant S), default and merged trunk -> gpurun_out/r3_train/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3_train
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for tag in default merged; do
  [ $tag = merged ] && export DANA_MERGE_TRUNK=1 || unset DANA_MERGE_TRUNK
  rm -rf /tmp/rp_$tag
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$tag -o bench -- python $R/bench.py --launch eager --mode step --steps 10 --warmup 5 --no-cpu-baseline > $O/$tag.log 2>&1
  cp $(find /tmp/rp_$tag -name "*kernel_stats.csv" | head -1) $O/${tag}_kernel_stats.csv
  tail -1 $O/$tag.log | cut -c1-260
done
