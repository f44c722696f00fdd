#!/bin/bash
# GPU box: kernel trace of the c3 step (eager launches, serial streams), per launch shape -> gpurun_out/c3_by_shape.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd $R
OUT=$R/gpurun_out/c3_trace
mkdir -p $OUT
COMMON="--serial_streams --cpu_baseline_s 0 --no_config5 --no_reg_only --sweep none --no_sensors --no_allreduce_rehearsal --warmup_s 0 --steps 4 --warmup 2 --no_kernel_events"
rocprofv3 --kernel-trace -d $OUT -o run --output-format csv -- python3 bench.py $COMMON > $OUT/bench.json 2> $OUT/bench.err
cd tools && python3 trace_by_shape.py $OUT/run_kernel_trace.csv 6 0.05 > $R/gpurun_out/c3_by_shape.txt
rm -f $OUT/run_kernel_trace.csv $OUT/*agent_info*
tail -c 300 $OUT/bench.err
