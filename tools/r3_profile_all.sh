#!/bin/bash
# rocprofv3 evidence for profiles/r3_*: kernel stats (default two-stream eager, hipGraph replay, single-stream, merged
# trunk single-stream, training iteration), the PMC passes (FETCH_SIZE / WRITE_SIZE / MFMA busy), the bench lines.
# Run on the GPU box: bash tools/r3_profile_all.sh   -> gpurun_out/prof_r3/
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run_stats() {  # name, bench args...
  name=$1; shift
  rm -rf /tmp/rp_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$name -o bench -- python $R/bench.py "$@" > $O/$name.log 2>&1
  f=$(find /tmp/rp_$name -name "*kernel_stats.csv" | head -1)
  cp $f $O/${name}_kernel_stats.csv
  tail -1 $O/$name.log | cut -c1-200
}
COMMON="--steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-train-step --no-secondary --no-pmc"
run_stats default --launch eager $COMMON
run_stats graph --launch graph $COMMON
run_stats single_stream --launch eager $COMMON --single-stream
DANA_MERGE_TRUNK=1 run_stats merged_single_stream --launch eager $COMMON --single-stream
run_stats train_step --launch eager --mode step --steps 10 --warmup 5 --no-cpu-baseline
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o pmc -- python $R/bench.py --launch eager --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-train-step --no-secondary --no-pmc --single-stream > $O/pmc_$c.log 2>&1
  cp $(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1) /tmp/pmc_$c.csv
done
python $R/tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE.csv /tmp/pmc_WRITE_SIZE.csv 4 $O/pmc_traffic.json > $O/pmc_traffic.txt
rm -rf /tmp/pmc_mfma
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_mfma -o pmc -- python $R/bench.py --launch eager --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-train-step --no-secondary --no-pmc --single-stream > $O/pmc_mfma.log 2>&1
python $R/tools/r3_secondary.py $O/single_stream_kernel_stats.csv $O/pmc_traffic.json $(find /tmp/pmc_mfma -name "*counter_collection.csv" | head -1) $O/secondary_rooflines.md "BASELINE configs[2] (600x1000 queries, way 2, shot 3, 4 episodes), train-mode forward" 4 600 1000 3 512 | tail -12
cd $R && python bench.py --dump-launches $O/launches.txt > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
python bench.py --mode step --batch 1 --no-cpu-baseline > $O/bench_step_b1.json 2>/dev/null
python bench.py --mode eval --batch 1 --no-cpu-baseline --no-pmc > $O/bench_eval_b1.json 2>/dev/null; cut -c1-200 $O/bench_eval_b1.json
