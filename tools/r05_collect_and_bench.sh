#!/bin/bash
# GPU box, end of round 5: rocprofv3 passes over both bench configurations (tools/collect_profiles.sh), the c5 passes once more on the round-4 form of the
# 16-bit path (L2I_H8_IMG_CONVS / RGB_FUSED / MOD_MULTI = 0: same-day yardstick for the byte and launch counts), then the driver's command three times.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
bash tools/collect_profiles.sh c3 > gpurun_out/r05_collect_c3.log 2>&1
bash tools/collect_profiles.sh c5 > gpurun_out/r05_collect_c5.log 2>&1
( export L2I_H8_IMG_CONVS=0 L2I_H8_RGB_FUSED=0 L2I_H8_MOD_MULTI=0 L2I_ROUND=r05old; bash tools/collect_profiles.sh c5 > gpurun_out/r05_collect_c5_old.log 2>&1 )
tail -3 gpurun_out/r05_collect_c3.log gpurun_out/r05_collect_c5.log
for t in a b c; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_c3_driver_cmd_$t.json 2> gpurun_out/r05_bench_c3_driver_cmd_$t.err
done
python3 - <<'PY'
import json,glob,os,csv
R=os.environ.get('GRAFT_REPO_ROOT','.')
for f in sorted(glob.glob(R+'/gpurun_out/r05_bench_c3_driver_cmd_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), d['value'], d['ms_per_step'], d['median_ms'], d['min_ms'], d['max_ms'], 'traffic', d['roofline']['traffic'], 'frac', d['roofline']['frac'], 'c5', d['config5']['value'], d['config5']['roofline']['traffic'], 'reg', {k:v['value'] for k,v in d['reg_only'].items()}, d['sensors']['during_timed'][:2], d['bench_wall_s'])
for tag in ('r05_c3', 'r05_c5', 'r05old_c5'):
    t=json.load(open('%s/gpurun_out/%s/hbm_traffic.json' % (R, tag)))
    n=sum(int(r['Calls']) for r in csv.DictReader(open('%s/gpurun_out/%s/stats/run_kernel_stats.csv' % (R, tag))))
    print(tag, 'GB/step', t['all_kernels_GB_per_step'], 'launches/step', round(n/13.0,1), 'src', t.get('src_sha256_16'))
PY
