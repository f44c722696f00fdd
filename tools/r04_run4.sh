#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_run4
mkdir -p $OUT
cd $R
timeout 300 python tools/probes/wino4_bench.py 8 > $OUT/wino4_bench.txt 2>&1
tail -12 $OUT/wino4_bench.txt
for rs in 2 3; do L2I_W4_RS=$rs python3 tools/probes/one_wino4.py 512 512 64 8; L2I_W4_RS=$rs python3 tools/probes/one_wino4.py 256 256 128 8; L2I_W4_RS=$rs python3 tools/probes/one_wino4.py 128 128 256 8; L2I_W4_RS=$rs python3 tools/probes/one_wino4.py 64 64 1024 8; done 2>&1 | grep -v amdgpu.ids | tee $OUT/w4_rs.txt
timeout 1400 python -m pytest tests -q -m gpu --durations=30 > $OUT/pytest.log 2>&1
tail -70 $OUT/pytest.log | cut -c1-220
