#!/bin/bash
# kernel times and launch timeline of one proposal-layer sort under rocprofv3. usage: sort_profile.sh B n topn
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
rm -rf /tmp/sp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -o s -- python $R/tools/one_sort.py "$@" > /tmp/sp.log 2>&1
F=$(find /tmp/sp -name "*kernel_stats.csv" | head -1)
T=$(find /tmp/sp -name "*kernel_trace.csv" | head -1)
[ -n "$F" ] && head -8 "$F" < /dev/null
[ -n "$T" ] && python - "$T" <<PY
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "ss_" in r["Kernel_Name"] or "topk" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[-8]["Start_Timestamp"]) if len(rows) >= 8 else 0
for r in rows[-8:]:
    print("%-28s start %7d ns  dur %6d ns" % (r["Kernel_Name"][:28], int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
PY
