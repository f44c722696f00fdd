#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 passes over bench.py for one config, outputs under gpurun_out/<tag>/.
#   bash tools/collect_profiles.sh c3|c5
# Counters in their own passes (never with --sys-trace & co.); the program itself follows `--` (no env / bash -c hop).
set -u
CFG=${1:-c3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${L2I_ROUND:-r06}_$CFG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
EXTRA=""
if [ "$CFG" = "c5" ]; then EXTRA="--config c5 --hip_graph 0"; fi
COMMON="--serial_streams --cpu_baseline_s 0 --no_config5 --no_reg_only --sweep none --no_sensors --no_allreduce_rehearsal --warmup_s 0 --steps 5 --warmup 2"
# 1. kernel trace + stats: 2 warm-up + 5 timed + (1 untimed + 5) steps with per-launch events = 13 steps in the file
rocprofv3 --kernel-trace --stats -d $OUT/stats -o run --output-format csv -- python3 bench.py $EXTRA $COMMON > $OUT/bench_serial_streams.json 2> $OUT/bench_serial_streams.err
# 2. HBM traffic: FETCH_SIZE and WRITE_SIZE do not fit one pass
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_f -o run --output-format csv -- python3 bench.py $EXTRA $COMMON --steps 1 --warmup 1 --no_kernel_events > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_w -o run --output-format csv -- python3 bench.py $EXTRA $COMMON --steps 1 --warmup 1 --no_kernel_events > /dev/null 2>&1
python3 tools/hbm_traffic.py $OUT/pmc_f $OUT/pmc_w $OUT/hbm_traffic.json $CFG > $OUT/hbm_traffic.log 2>&1
# 3. SQ counters of one step per kernel
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAVE_CYCLES -d $OUT/pmc_sq -o run --output-format csv -- python3 bench.py $EXTRA $COMMON --steps 1 --warmup 1 --no_kernel_events > /dev/null 2>&1
python3 tools/sq_by_kernel.py $OUT/pmc_sq/run_counter_collection.csv $OUT/step_sq_counters_by_kernel.csv
python3 tools/kernel_table.py $OUT/stats/run_kernel_stats.csv $OUT/hbm_traffic.json $OUT/step_sq_counters_by_kernel.csv $OUT/kernel_table.md 13 > /dev/null 2>&1      # 2 warm-up + 5 timed + (1 untimed + 5) event-pass steps
rm -rf $OUT/pmc_f $OUT/pmc_w $OUT/pmc_sq/*agent_info* 2>/dev/null
ls -la $OUT $OUT/stats | head -30
tail -c 400 $OUT/bench_serial_streams.err
