export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; rm -rf /tmp/et
rocprofv3 --kernel-trace --output-format csv -d /tmp/et -o t -- python $R/bench.py --launch program --mode train --steps 6 --warmup 3 \
  --no-cpu-baseline --no-roofline --no-train-step --no-secondary --no-pmc > /tmp/et.log 2>&1
T=$(find /tmp/et -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_gaps.py $T rcnn_loss_c
