#!/bin/bash
# RoIAlign forward alone under rocprofv3: where do its tap loads go? One --pmc group per run (--kernel-trace only), the
# counters of the vector-memory path (TCP = per-CU L1, TCC = L2) and of the instruction mix, for tools/roi_bench.py's 55
# launches of roi_align_fwd_nhwc on the bench's own proposals. Output: gpurun_out/roi_prof/roi_counters.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/roi_prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/tools/roi_bench.py > $O/roi_bench.txt 2>$O/roi_bench.err; cat $O/roi_bench.txt
rocprofv3 --list-avail > $O/avail.txt 2>&1 || rocprofv3 -L > $O/avail.txt 2>&1
grep -o "TCP_[A-Z0-9_]*\|TCC_[A-Z0-9_]*\|TA_[A-Z0-9_]*" $O/avail.txt | sort -u | tr '\n' ' ' | cut -c1-3000 > $O/avail_mem_counters.txt
: > $O/roi_counters.txt
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE" \
           "TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rm -rf /tmp/roi_$tag
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/roi_$tag -o pmc -- python $R/tools/roi_bench.py > /tmp/roi_$tag.log 2>&1
  f=$(find /tmp/roi_$tag -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then
    python - "$f" >> $O/roi_counters.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if "roi_align_fwd_nhwc" in row.get("Kernel_Name", ""):
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    print("%-34s per launch: %.4g  (%d launches)" % (k, sum(v) / len(v), len(v)))
PY
  else
    echo "group [$grp]: no counter file: $(grep -i "error\|invalid\|not" /tmp/roi_$tag.log | head -2)" >> $O/roi_counters.txt
  fi
done
cat $O/roi_counters.txt
