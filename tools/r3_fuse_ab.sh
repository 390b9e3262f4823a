#!/bin/bash
# A/B of the fused bottleneck tail (DANA_FUSE_TAIL) on one box -> gpurun_out/r3_fuse/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3_fuse
mkdir -p $O
cd $R
for v in 1 0 1 0; do
  DANA_FUSE_TAIL=$v python bench.py --steps 60 --warmup 10 --no-pmc --no-cpu-baseline --no-secondary --no-train-step --dump-launches $O/launches_$v.txt > $O/f_$v.json 2> $O/f_$v.err
  python - <<PY
import json
try:
    j=json.loads(open("$O/f_$v.json").read().strip().splitlines()[-1])
    print("fuse $v forward", j["value"], j["ms_per_step"], j["launch"] if "launch" in j else "", "frac", j["roofline"]["frac"], "launches", j["roofline"].get("launches_per_step"), "kernel_ms", j["roofline"].get("kernel_ms_per_step"))
except Exception as e:
    print("fuse $v failed", e)
PY
done
