#!/bin/bash
# SQ counter passes over single launches of the h8 conv kernel (gpurun): where do the wave cycles go?   usage: bash tools/probes/h8_counters.sh ["cin cout k stride res" ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/h8_ctr
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
if [ $# -eq 0 ]; then set -- "64 64 3 1 1024" "128 128 3 1 512" "512 512 3 1 64" "64 256 1 1 256"; fi
for shape in "$@"; do
  tag=$(echo $shape | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA -d $OUT/a_$tag -o run --output-format csv -- python3 tools/probes/one_h8.py $shape > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU -d $OUT/b_$tag -o run --output-format csv -- python3 tools/probes/one_h8.py $shape > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT -d $OUT/c_$tag -o run --output-format csv -- python3 tools/probes/one_h8.py $shape > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get('GRAFT_REPO_ROOT', os.getcwd()) + '/gpurun_out/h8_ctr'
for d in sorted(glob.glob(out + '/*_*')):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        name = ''
        for r in csv.DictReader(open(f)):
            if 'conv_h8' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value'])); name = r['Kernel_Name'][:60]
        print(os.path.basename(d), name, {k: '%.4g' % (sum(v) / len(v)) for k, v in acc.items()})
PY
rm -rf $OUT/*/ 2>/dev/null
