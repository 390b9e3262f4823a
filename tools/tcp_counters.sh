#!/bin/bash
# L1 (TCP) / L2 (TCC) request counters of the split igemm on one shape: bash tools/tcp_counters.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace -d /tmp/tcp$i -o c --output-format csv -- python $R/tools/one_conv.py ${SHAPE:-512 4 4 512 2048 1 1 0 3} > /dev/null 2>&1
  f=$(find /tmp/tcp$i -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys
agg = {}
for r in csv.DictReader(open(sys.argv[1])):
    if "igemm" in r["Kernel_Name"]:
        agg[r["Counter_Name"]] = agg.get(r["Counter_Name"], 0) + float(r["Counter_Value"])
for k, v in sorted(agg.items()):
    print("  %-36s %16.0f" % (k, v))
PY
done
