#!/bin/bash
# per-kernel times of the proposal layer's NMS pair under rocprofv3 (tools/nms_bench.py). usage: nms_profile.sh
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
rm -rf /tmp/np
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/np -o s -- python $R/tools/nms_bench.py > /tmp/np.log 2>&1
tail -1 /tmp/np.log
T=$(find /tmp/np -name "*kernel_trace.csv" | head -1)
[ -n "$T" ] && python - "$T" <<PY
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "nms_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[-8]["Start_Timestamp"])
for r in rows[-8:]:
    print("%-20s start %7d ns  dur %6d ns" % (r["Kernel_Name"].split("::")[-1][:20], int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
PY
