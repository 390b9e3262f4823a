#!/bin/bash
# rocprofv3 evidence for profiles/: kernel stats (default multi-stream, single-stream, training iteration) and the two
# PMC passes (FETCH_SIZE / WRITE_SIZE) -> gpurun_out/prof_final/. Run on the GPU box: bash tools/profile_all.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/prof_r2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run_stats() {  # name, bench args...
  name=$1; shift
  rm -rf /tmp/rp_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$name -o bench -- python $R/bench.py "$@" > $O/$name.log 2>&1
  f=$(find /tmp/rp_$name -name "*kernel_stats.csv" | head -1)
  cp $f $O/${name}_kernel_stats.csv
  tail -1 $O/$name.log | cut -c1-300
}
run_stats default --launch eager --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-train-step --no-secondary --no-pmc
run_stats graph --launch graph --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-train-step --no-secondary --no-pmc
run_stats single_stream --launch eager --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-train-step --no-secondary --no-pmc --single-stream
run_stats train_step --launch eager --mode step --steps 10 --warmup 5 --no-cpu-baseline
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o pmc -- python $R/bench.py --launch eager --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-train-step --no-secondary --no-pmc --single-stream > $O/pmc_$c.log 2>&1
  cp $(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1) /tmp/pmc_$c.csv
done
python $R/tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE.csv /tmp/pmc_WRITE_SIZE.csv 4 $O/pmc_traffic.json | tee $O/pmc_traffic.txt
rm -rf /tmp/pmc_mfma
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_mfma -o pmc -- python $R/bench.py --launch eager --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-train-step --no-secondary --no-pmc --single-stream > $O/pmc_mfma.log 2>&1
python $R/tools/secondary_rooflines.py $O/single_stream_kernel_stats.csv $O/pmc_traffic.json $(find /tmp/pmc_mfma -name "*counter_collection.csv" | head -1) 13 $O/secondary_rooflines.md | tail -30
cd $R && python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-400 $O/bench.json
python tools/gemm_power.py $O/gemm_power.md > $O/gemm_power.log 2>&1; tail -12 $O/gemm_power.log
python tools/tile_sweep.py $O/tile_sweep.md > /dev/null 2>&1
python bench.py --mode step --batch 1 --no-cpu-baseline > $O/bench_step_b1.json 2>/dev/null
python bench.py --height 800 --width 1333 --shot 10 --batch 2 --no-cpu-baseline --no-pmc > $O/bench_cfg4.json 2>/dev/null; cut -c1-300 $O/bench_cfg4.json
