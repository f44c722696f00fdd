#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_run3
mkdir -p $OUT
cd $R
python3 tools/probes/gc_cycles.py 64 > $OUT/gc_cycles.txt 2>&1
tail -30 $OUT/gc_cycles.txt
for shape in "512 512 64 8" "64 64 1024 8" "128 128 256 8"; do
  echo "== $shape" >> $OUT/w4_ablations.txt
  python3 tools/probes/one_wino4.py $shape off >> $OUT/w4_ablations.txt 2>&1
  python3 tools/probes/one_wino4.py $shape all >> $OUT/w4_ablations.txt 2>&1
  for v in DMA XF MFMA LDSD EPI XF+LDSD DMA+XF+LDSD DMA+XF+LDSD+EPI MFMA+XF+LDSD; do
    L2I_LIB=$R/tools/ab/libl2i_w4_no_$v.so python3 tools/probes/one_wino4.py $shape all >> $OUT/w4_ablations.txt 2>&1
  done
done
grep -v amdgpu.ids $OUT/w4_ablations.txt
cd /tmp && export TMPDIR=/tmp
cd $R
for shape in "512 512 64 8" "64 64 1024 8"; do
  tag=$(echo $shape | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA -d $OUT/a_$tag -o run --output-format csv -- python3 tools/probes/one_wino4.py $shape all plain 4 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU -d $OUT/b_$tag -o run --output-format csv -- python3 tools/probes/one_wino4.py $shape all plain 4 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM SQ_WAVES SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_VMEM -d $OUT/c_$tag -o run --output-format csv -- python3 tools/probes/one_wino4.py $shape all plain 4 > /dev/null 2>&1
done
python3 - <<'PY' > $OUT/w4_counters.txt
import csv, glob, os, collections
out = os.environ.get('GRAFT_REPO_ROOT', os.getcwd()) + '/gpurun_out/r04_run3'
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
