#!/bin/bash
# Runs on the GPU box: the driver's exact bench command three times (a, b, c) + two lean diagnostic runs (host run-ahead unbounded / bounded,
# count-only warm-up as in round 3) so that per-step times, clocks and allocator growth can be compared.  Outputs under gpurun_out/r04_repro/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_repro
mkdir -p $OUT
cd $R
ls /sys/class/drm/ > $OUT/sysfs.txt 2>&1
for d in /sys/class/drm/card*/device; do echo "== $d"; ls $d | tr '\n' ' '; echo; cat $d/pp_dpm_sclk 2>&1 | head -5; ls $d/hwmon/* 2>&1 | tr '\n' ' '; echo; cat $d/hwmon/hwmon*/power1_* 2>&1 | head; done >> $OUT/sysfs.txt 2>&1
LEAN="--cpu_baseline_s 0 --no_config5 --no_reg_only --sweep none --no_kernel_events"
python3 bench.py --gpus 1 --steps 20 --warmup 5 --max_ahead 0 --warmup_s 0 $LEAN > $OUT/lean_r03_conditions.json 2> $OUT/lean_r03_conditions.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 --max_ahead 2 --warmup_s 0 $LEAN > $OUT/lean_ahead2.json 2> $OUT/lean_ahead2.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 $LEAN > $OUT/lean_default.json 2> $OUT/lean_default.err
for t in a b c; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/driver_cmd_$t.json 2> $OUT/driver_cmd_$t.err
done
tail -c 600 $OUT/*.err
python3 - <<'PY'
import json,glob,os
for f in sorted(glob.glob(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r04_repro/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d['value'], d['ms_per_step'], d['median_ms'], d['min_ms'], d['max_ms'], d['allocator'], d['sensors']['before_timed'], d['sensors']['after_timed'], d.get('bench_wall_s'))
        print('   ', d['step_ms'])
        if d.get('config5'): print('    c5', d['config5']['value'], d['config5']['median_ms'], 'reg_only', {k:v['value'] for k,v in (d.get('reg_only') or {}).items()}, 'allreduce_us', d['allreduce_us'])
    except Exception as e:
        print(f, 'ERR', e)
PY
