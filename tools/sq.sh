#!/bin/bash
# SQ counters of one split-kernel shape: bash tools/sq.sh "<one_conv args>" tag
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
ARGS=$1; TAG=$2
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_MISC"; do
  rm -rf /tmp/sq_$TAG
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/sq_$TAG -o pmc -- python $R/tools/one_conv.py $ARGS 6 > /tmp/sq_$TAG.log 2>&1
  f=$(find /tmp/sq_$TAG -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/tools/sq_report.py $f igemm_split || tail -5 /tmp/sq_$TAG.log
done
