#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_run10
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_networks_gpu.py -q -m gpu -k "winograd or vgg or content or perceptual or full_size_1024_forward" > $OUT/pytest.log 2>&1
grep -E "passed|failed|Error|assert " $OUT/pytest.log | cut -c1-300 | tail -10
for shape in "64 64 1024 8" "128 128 512 8"; do python3 tools/probes/one_wino4.py $shape all relu_in; python3 tools/probes/one_wino4.py $shape all plain; python3 tools/probes/one_wino4.py $shape off relu_in; done 2>&1 | grep -v amdgpu
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no_config5 --no_reg_only --sweep none --cpu_baseline_s 0 > $OUT/bench.json 2> $OUT/bench.err
python3 - <<'PY'
import json,os
R=os.environ.get('GRAFT_REPO_ROOT','.')
d=json.loads(open(R+'/gpurun_out/r04_run10/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['median_ms'], d['min_ms'], d['max_ms'], [(f['family'],f['launches_per_step'],f['ms_per_step'],f['frac']) for f in d['roofline']['families']][:3])
PY
