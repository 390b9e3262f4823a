#!/bin/bash
# SQ counters of the split igemm on the RPN conv shape: bash tools/sq_split.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export DANA_MFMA_SPLIT=${DANA_MFMA_SPLIT:-1}
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_LEVEL_LDS"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d /tmp/sq$i -o sq --output-format csv -- python $R/tools/one_conv.py ${SHAPE:-4 38 63 2048 512 3 1 0 3} > /dev/null 2>&1
  f=$(find /tmp/sq$i -name "*counter_collection.csv" | head -1)
  python $R/tools/sq_report.py $f igemm | grep -v "^$"
done
