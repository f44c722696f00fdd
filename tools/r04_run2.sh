#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_run2
mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "winograd" > $OUT/pytest_wino.log 2>&1
tail -15 $OUT/pytest_wino.log
timeout 300 python tools/probes/wino4_bench.py 8 > $OUT/wino4_bench.txt 2>&1
cat $OUT/wino4_bench.txt | tail -20
LEAN="--cpu_baseline_s 0 --no_config5 --no_reg_only --sweep none --no_kernel_events"
L2I_WINO4=off python3 bench.py --steps 60 --warmup 5 --no_sensors $LEAN > $OUT/lean_nosens_60.json 2> $OUT/e1
L2I_WINO4=off python3 bench.py --steps 60 --warmup 5 --no_sensors --no_gc $LEAN > $OUT/lean_nosens_nogc_60.json 2> $OUT/e2
L2I_WINO4=off python3 bench.py --steps 60 --warmup 5 $LEAN > $OUT/lean_sens_60.json 2> $OUT/e3
python3 bench.py --steps 20 --warmup 5 --no_sensors $LEAN > $OUT/lean_wino4.json 2> $OUT/e4
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04_run2/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d['value'], d['ms_per_step'], d['median_ms'], d['min_ms'], d['max_ms'], d['allocator'])
        print('   ', d['step_ms'])
    except Exception as e:
        print(f, 'ERR', e)
PY
tail -3 $OUT/e4
