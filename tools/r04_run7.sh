#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_run7
mkdir -p $OUT
cd $R
(free -g; cat /sys/fs/cgroup/memory.max 2>/dev/null; cat /sys/fs/cgroup/memory/memory.limit_in_bytes 2>/dev/null; nproc) > $OUT/host.txt 2>&1
cat $OUT/host.txt
timeout 1500 python -m pytest tests/test_networks_gpu.py tests/test_h8_gpu.py -q -m gpu -k "config3_whole or config4_whole or walk_gradient_1024 or regressor or resnet or bf16_networks or fp32_trunk or training_step_vs" --durations=12 > $OUT/pytest.log 2>&1
grep -E "passed|failed|Error|assert |^[0-9.]+s " $OUT/pytest.log | cut -c1-300 | tail -30
