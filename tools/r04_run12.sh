#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_run12
mkdir -p $OUT
cd $R
python3 tools/probes/torgb_bench.py 2>&1 | grep -v amdgpu | tee $OUT/torgb_bench.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_h8_gpu.py tests/test_networks_gpu.py tests/test_trajectory_gpu.py -q -m gpu -k "torgb or generator or bit_stable or full_size_1024_forward or h8_torgb or bf16_networks" > $OUT/pytest.log 2>&1
grep -E "passed|failed|Error|assert " $OUT/pytest.log | cut -c1-300 | tail -8
