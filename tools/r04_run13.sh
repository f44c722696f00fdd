#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_run13
mkdir -p $OUT
cd $R
(cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpuset.cpus.effective 2>/dev/null; nproc; python3 -c "import os; print(len(os.sched_getaffinity(0)))") > $OUT/host_cpu.txt 2>&1
cat $OUT/host_cpu.txt
python3 bench.py --config c5 --steps 20 --warmup 5 > $OUT/r04_bench_c5.json 2> $OUT/c5.err
python3 bench.py --config c5 --hip_graph 0 --steps 20 --warmup 5 --cpu_baseline_s 0 --no_reg_only --sweep none --no_kernel_events > $OUT/r04_bench_c5_eager_launches.json 2> $OUT/c5e.err
python3 - <<'PY'
import json,os
R=os.environ.get('GRAFT_REPO_ROOT','.')
for f in ('r04_bench_c5.json','r04_bench_c5_eager_launches.json'):
    d=json.loads(open(R+'/gpurun_out/r04_run13/'+f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['median_ms'], d['min_ms'], d['max_ms'], d['dtype'], d['config']['hip_graph'], (d.get('roofline') or {}).get('traffic'), d.get('batch_sweep') and [(b['batch'],b['images_s']) for b in d['batch_sweep']], d.get('reg_only') and {k:v['value'] for k,v in d['reg_only'].items()})
PY
cd /tmp && export TMPDIR=/tmp
cd $R
for shape in "512 512 64 8 all plain" "64 64 1024 8 all relu_in"; do
  tag=$(echo $shape | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA -d $OUT/a_$tag -o run --output-format csv -- python3 tools/probes/one_wino4.py $shape 4 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU -d $OUT/b_$tag -o run --output-format csv -- python3 tools/probes/one_wino4.py $shape 4 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_WAVES SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_VMEM -d $OUT/c_$tag -o run --output-format csv -- python3 tools/probes/one_wino4.py $shape 4 > /dev/null 2>&1
done
python3 - <<'PY' > $OUT/w4_counters.txt
import csv, glob, os, collections
out = os.environ.get('GRAFT_REPO_ROOT', os.getcwd()) + '/gpurun_out/r04_run13'
for d in sorted(glob.glob(out + '/[abc]_*')):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'conv_wino4' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        print(os.path.basename(d), {k: '%.4g' % (sum(v) / len(v)) for k, v in acc.items()})
PY
cat $OUT/w4_counters.txt
rm -rf $OUT/[abc]_*/ 2>/dev/null
