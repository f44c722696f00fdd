#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/collect_profiles.sh c3 > gpurun_out/r05_collect_c3.log 2>&1
bash tools/collect_profiles.sh c5 > gpurun_out/r05_collect_c5.log 2>&1
tail -5 gpurun_out/r05_collect_c3.log gpurun_out/r05_collect_c5.log
cat gpurun_out/r05_c3/hbm_traffic.log | tail -12
for t in a b c; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_c3_driver_cmd_$t.json 2> gpurun_out/r05_bench_c3_driver_cmd_$t.err
done
python3 - <<'PY'
import json,glob,os
R=os.environ.get('GRAFT_REPO_ROOT','.')
for f in sorted(glob.glob(R+'/gpurun_out/r05_bench_c3_driver_cmd_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), d['value'], d['ms_per_step'], d['median_ms'], d['min_ms'], d['max_ms'], 'traffic', d['roofline']['traffic'], 'frac', d['roofline']['frac'], 'c5', d['config5']['value'], d['config5']['roofline']['traffic'], 'reg', {k:v['value'] for k,v in d['reg_only'].items()}, d['sensors']['during_timed'][:2], d['bench_wall_s'])
PY
