#!/bin/bash
# GPU box: WRITE_SIZE of the c5 step per launch shape, new path and (L2I_H8_*=0) the round-4 form
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
cd $R
COMMON="--config c5 --hip_graph 0 --serial_streams --cpu_baseline_s 0 --no_config5 --no_reg_only --sweep none --no_sensors --no_allreduce_rehearsal --warmup_s 0 --steps 1 --warmup 1 --no_kernel_events"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_w_new -o run --output-format csv -- python3 bench.py $COMMON > /dev/null 2>&1
python3 tools/probes/pmc_by_shape.py $R/gpurun_out/pmc_w_new/run_counter_collection.csv WRITE_SIZE 2 > $R/gpurun_out/c5_write_new.txt
export L2I_H8_IMG_CONVS=0 L2I_H8_RGB_FUSED=0 L2I_H8_MOD_MULTI=0
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_w_old -o run --output-format csv -- python3 bench.py $COMMON > /dev/null 2>&1
python3 tools/probes/pmc_by_shape.py $R/gpurun_out/pmc_w_old/run_counter_collection.csv WRITE_SIZE 2 > $R/gpurun_out/c5_write_old.txt
rm -rf $R/gpurun_out/pmc_w_new $R/gpurun_out/pmc_w_old
