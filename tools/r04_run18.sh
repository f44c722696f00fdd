#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r04_run18
cd $R
timeout 900 python -m pytest tests/test_h8_gpu.py tests/test_trajectory_gpu.py -q -m gpu > gpurun_out/r04_run18/pytest.log 2>&1
tail -3 gpurun_out/r04_run18/pytest.log
bash tools/r04_run8.sh
