#!/bin/bash
# GPU box, end of round 6: rocprofv3 passes over both bench configurations (tools/collect_profiles.sh: kernel trace + stats, FETCH_SIZE / WRITE_SIZE passes,
# one SQ-counter pass), the per-shape traces, then the driver's command three times.  Outputs under gpurun_out/r06_*; tools/publish_profiles.sh r06 copies
# the summaries into profiles/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export L2I_ROUND=r06
bash tools/collect_profiles.sh c3 > gpurun_out/r06_collect_c3.log 2>&1
bash tools/collect_profiles.sh c5 > gpurun_out/r06_collect_c5.log 2>&1
tail -3 gpurun_out/r06_collect_c3.log gpurun_out/r06_collect_c5.log
python3 tools/trace_by_shape.py gpurun_out/r06_c3/stats/run_kernel_trace.csv 13 > gpurun_out/r06_c3_by_shape.txt 2>/dev/null
python3 tools/trace_by_shape.py gpurun_out/r06_c5/stats/run_kernel_trace.csv 13 > gpurun_out/r06_c5_by_shape.txt 2>/dev/null
for t in a b c; do
  python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06_bench_c3_driver_cmd_$t.json 2> gpurun_out/r06_bench_c3_driver_cmd_$t.err
done
python3 bench.py --config c5 --steps 20 --warmup 5 --cpu_baseline_s 0 --sweep none > gpurun_out/r06_bench_c5.json 2> gpurun_out/r06_bench_c5.err
python3 - <<'PY'
import json,glob,os,csv
R=os.environ.get('GRAFT_REPO_ROOT','.')
for f in sorted(glob.glob(R+'/gpurun_out/r06_bench_c3_driver_cmd_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), d['value'], d['ms_per_step'], d['median_ms'], d['min_ms'], d['max_ms'], 'traffic', d['roofline']['traffic'], 'frac', d['roofline']['frac'], 'c5', d['config5']['value'], d['config5']['roofline']['traffic'], 'reg', {k:v['value'] for k,v in d['reg_only'].items()}, d['sensors']['during_timed'][:2], d['bench_wall_s'])
d=json.loads(open(R+'/gpurun_out/r06_bench_c5.json').read().strip().splitlines()[-1]); print('c5 isolated', d['value'], d['ms_per_step'])
for tag in ('r06_c3', 'r06_c5'):
    t=json.load(open('%s/gpurun_out/%s/hbm_traffic.json' % (R, tag)))
    n=sum(int(r['Calls']) for r in csv.DictReader(open('%s/gpurun_out/%s/stats/run_kernel_stats.csv' % (R, tag))))
    print(tag, 'GB/step', t['all_kernels_GB_per_step'], 'launches/step', round(n/13.0,1), 'src', t.get('src_sha256_16'))
PY
