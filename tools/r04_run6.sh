#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_run6
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests/test_trajectory_gpu.py tests/test_h8_gpu.py tests/test_kernels_gpu.py tests/test_networks_gpu.py -q -m gpu -k "trajectory or bf16_networks or fp32_trunk or difference_residual or forward_parity_and_consistency or bf16_training_step or winograd" -s --durations=10 > $OUT/pytest_new.log 2>&1
grep -E "trajectory |passed|failed|Error|assert|^\{'case" $OUT/pytest_new.log | cut -c1-900 | tail -30
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_a.json 2> $OUT/bench_a.err
python3 - <<'PY'
import json,os
R=os.environ.get('GRAFT_REPO_ROOT','.')
d=json.loads(open(R+'/gpurun_out/r04_run6/bench_a.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['median_ms'], d['min_ms'], d['max_ms'], d['allocator'], d['bench_wall_s'])
print(d['step_ms'])
print([(f['family'], f['launches_per_step'], f['ms_per_step'], f['frac']) for f in d['roofline']['families']])
print('c5', d['config5']['value'], d['config5']['median_ms'], d['config5']['step_ms'], d['config5']['roofline']['frac'], 'reg_only', {k:(v['value'],v['median_ms']) for k,v in d['reg_only'].items()}, 'allreduce', d['allreduce_us'])
print(d['sensors']['before_timed'], d['sensors']['after_timed'])
PY
tail -3 $OUT/bench_a.err
