#!/bin/bash
# The profiles/<tag>_* evidence set in one call on the GPU box (replaces the per-round profile scripts):
#   bash tools/profile_all.sh r4                  -> gpurun_out/prof_r4/       BASELINE configs[2] (the bench default)
#   bash tools/profile_all.sh r4_cfg4 cfg4        -> gpurun_out/prof_r4_cfg4/  configs[4]'s per-GPU shape (800x1333, shot 10, 2 episodes)
# rocprofv3 kernel stats (two-stream eager, hipGraph replay, single stream = every kernel alone, training iteration), the PMC
# passes (FETCH_SIZE / WRITE_SIZE / MFMA busy, one counter group per run, --kernel-trace only), secondary rooflines, the
# bench lines with the per-launch tables. Copy what should be judged into profiles/.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r4}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CFG=""; ROIS=512; H=600; W=1000; SHOT=3; B=4
WHAT="BASELINE configs[2] (600x1000 queries, way 2, shot 3, 4 episodes), train-mode forward"
if [ "$2" = cfg4 ]; then
  CFG="--height 800 --width 1333 --shot 10 --batch 2"; ROIS=256; H=800; W=1333; SHOT=10; B=2
  WHAT="BASELINE configs[4] per-GPU shape (800x1333 queries, way 2, shot 10, 2 episodes), train-mode forward"
fi
run_stats() {  # name, bench args...
  name=$1; shift
  rm -rf /tmp/rp_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp_$name -o bench -- python $R/bench.py $CFG "$@" > $O/$name.log 2>&1
  cp $(find /tmp/rp_$name -name "*kernel_stats.csv" | head -1) $O/${name}_kernel_stats.csv
  tail -1 $O/$name.log | cut -c1-200
}
COMMON="--steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-train-step --no-secondary --no-pmc"
run_stats default --launch eager $COMMON
run_stats single_stream --launch eager $COMMON --single-stream
if [ "$2" != cfg4 ]; then
  run_stats graph --launch graph $COMMON
  run_stats program --launch program $COMMON
  run_stats train_step --launch program --mode step --steps 10 --warmup 5 --no-cpu-baseline
fi
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o pmc -- python $R/bench.py $CFG --launch eager --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-train-step --no-secondary --no-pmc --single-stream > $O/pmc_$c.log 2>&1
  cp $(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1) /tmp/pmc_$c.csv
done
python $R/tools/pmc_traffic.py /tmp/pmc_FETCH_SIZE.csv /tmp/pmc_WRITE_SIZE.csv 4 $O/pmc_traffic.json > $O/pmc_traffic.txt
rm -rf /tmp/pmc_mfma
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_mfma -o pmc -- python $R/bench.py $CFG --launch eager --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-train-step --no-secondary --no-pmc --single-stream > $O/pmc_mfma.log 2>&1
cp $(find /tmp/pmc_mfma -name "*counter_collection.csv" | head -1) $O/pmc_mfma_counter_collection.csv
python $R/tools/secondary.py $O/single_stream_kernel_stats.csv $O/pmc_traffic.json $O/pmc_mfma_counter_collection.csv $O/secondary_rooflines.md "$WHAT" $B $H $W $SHOT $ROIS | tail -12
cd $R && python bench.py $CFG --dump-launches $O/launches.txt 2> $O/bench.err | grep '^{' > $O/bench.json; cut -c1-300 $O/bench.json  # (RCCL prints a banner on stdout)
if [ "$2" != cfg4 ]; then
  python tools/phase_times.py 30 > $O/phase_times.txt 2>&1
  python tools/roundtrip_gap.py 2>&1 | grep -v amdgpu.ids > $O/roundtrip_gap.txt
  python tools/program_hostprof.py 20 2>&1 | grep -v amdgpu.ids > $O/program_hostprof.txt
  python tools/stress_host_8proc.py 8 8 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" > $O/host_8proc.md
  bash tools/exposed_time.sh step program > $O/exposed_time_step.txt 2>&1
  bash tools/exposed_time.sh train program rcnn_loss_c > $O/exposed_time_forward.txt 2>&1
fi
