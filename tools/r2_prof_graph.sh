#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2e
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for L in graph eager; do
  rm -rf /tmp/rp_$L
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$L -o b -- python $R/bench.py --launch $L --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-train-step --no-secondary --no-pmc > $O/$L.log 2>&1
  cp $(find /tmp/rp_$L -name "*kernel_stats.csv" | head -1) $O/${L}_kernel_stats.csv
  cp $(find /tmp/rp_$L -name "*kernel_trace.csv" | head -1) $O/${L}_kernel_trace.csv
  tail -1 $O/$L.log | cut -c1-300
done
python - <<'PY'
import csv,os
O=os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/r2e'
for L in ('graph','eager'):
    rows=list(csv.DictReader(open('%s/%s_kernel_trace.csv'%(O,L))))
    rows.sort(key=lambda r:int(r['Start_Timestamp']))
    n=len(rows); rows=rows[n//2:]   # steady state second half
    t0=int(rows[0]['Start_Timestamp']); t1=max(int(r['End_Timestamp']) for r in rows)
    busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows)
    # union of intervals
    iv=sorted((int(r['Start_Timestamp']),int(r['End_Timestamp'])) for r in rows)
    u=0; cs,ce=iv[0]
    for s,e in iv[1:]:
        if s>ce: u+=ce-cs; cs,ce=s,e
        else: ce=max(ce,e)
    u+=ce-cs
    print(L,'kernels',len(rows),'span ms',(t1-t0)/1e6,'sum ms',busy/1e6,'union(busy) ms',u/1e6,'idle ms',(t1-t0-u)/1e6)
PY
gzip -f $O/*_kernel_trace.csv
