#!/bin/bash
# SQ counter passes over single Winograd launches (gpurun): where do the wave cycles go?
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/wino_ctr
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
export L2I_WINO_STAGGER=0
for shape in "64 64 1024 relu_in" "512 512 64 style" "64 64 1024 res"; do
  tag=$(echo $shape | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA -d $OUT/a_$tag -o run --output-format csv -- python3 tools/probes/one_wino_kind.py $shape > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU -d $OUT/b_$tag -o run --output-format csv -- python3 tools/probes/one_wino_kind.py $shape > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_WAVES SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_VMEM -d $OUT/c_$tag -o run --output-format csv -- python3 tools/probes/one_wino_kind.py $shape > /dev/null 2>&1
done
python3 - <<'PY'
import csv, glob, os, collections
out = os.environ.get('GRAFT_REPO_ROOT', os.getcwd()) + '/gpurun_out/wino_ctr'
for d in sorted(glob.glob(out + '/*_*')):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if 'conv_wino' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
        print(os.path.basename(d), {k: '%.4g' % (sum(v) / len(v)) for k, v in acc.items()})
PY
rm -rf $OUT/*/ 2>/dev/null
