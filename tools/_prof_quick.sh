R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_q
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_ss
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_ss -o bench -- python $R/bench.py --launch eager --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-train-step --no-secondary --no-pmc --single-stream > $O/ss.log 2>&1
cp $(find /tmp/rp_ss -name "*kernel_stats.csv" | head -1) $O/ss_kernel_stats.csv
tail -1 $O/ss.log | cut -c1-200
