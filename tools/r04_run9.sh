#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_run9
mkdir -p $OUT
cd $R
python3 tools/bf16_study.py 64,256,1024 > $OUT/r04_bf16_tolerance.json 2> $OUT/e1
python3 - <<'PY'
import json,os
R=os.environ.get('GRAFT_REPO_ROOT','.')
for l in open(R+'/gpurun_out/r04_run9/r04_bf16_tolerance.json'):
    if l.startswith('{'):
        d=json.loads(l)
        if d['case']=='networks': print('net',d['size'],{k:round(v,4) for k,v in d.items() if k.startswith('R16')})
        else: print('step',d['size'],d['attrs'],round(d['grad_cos'],4),round(d['grad_l2'],3))
PY
timeout 1500 python -m pytest tests/test_networks_gpu.py tests/test_h8_gpu.py -q -m gpu -s -k "config4_whole or data_parallel or regressor or resnet or full_size_training_step or bf16_networks" --durations=12 > $OUT/pytest.log 2>&1
grep -E "passed|failed|Error|assert |^[0-9.]+s |16-bit path" $OUT/pytest.log | cut -c1-300 | tail -30
bash tools/collect_profiles.sh c3 > gpurun_out/r04_collect_c3.log 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_final.json 2> $OUT/bench_final.err
python3 - <<'PY'
import json,os
R=os.environ.get('GRAFT_REPO_ROOT','.')
d=json.loads(open(R+'/gpurun_out/r04_run9/bench_final.json').read().strip().splitlines()[-1])
print(d['value'], d['median_ms'], d['min_ms'], d['max_ms'], 'traffic', d['roofline']['traffic'], d['roofline']['traffic_note'][:80], [(f['family'],f['launches_per_step'],f['ms_per_step'],f['frac']) for f in d['roofline']['families']])
PY
