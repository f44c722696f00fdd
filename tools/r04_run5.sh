#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_run5
mkdir -p $OUT
cd $R
for shape in "512 512 64 8" "64 64 1024 8"; do
  echo "== $shape" >> $OUT/w4_ablations.txt
  python3 tools/probes/one_wino4.py $shape all >> $OUT/w4_ablations.txt 2>&1
  for v in DMA XF DMA+XF+LDSD DMA+XF+LDSD+UREAD DMA+XF+LDSD+UREAD+BAR DMA+XF+LDSD+BAR BAR MFMA+XF+LDSD; do
    L2I_LIB=$R/tools/ab/libl2i_w4_no_$v.so python3 tools/probes/one_wino4.py $shape all >> $OUT/w4_ablations.txt 2>&1
  done
done
grep -v amdgpu.ids $OUT/w4_ablations.txt
L2I_TRUNK_F32=0 python3 tools/bf16_study.py 64,256 > $OUT/bf16_study_trunk_bf16.json 2> $OUT/e1
L2I_TRUNK_F32=1 python3 tools/bf16_study.py 64,256,1024 > $OUT/bf16_study_trunk_f32.json 2> $OUT/e2
python3 - <<'PY'
import json,os
R=os.environ.get('GRAFT_REPO_ROOT','.')
for f in ('bf16_study_trunk_bf16.json','bf16_study_trunk_f32.json'):
    print(f)
    for l in open(R+'/gpurun_out/r04_run5/'+f):
        if not l.startswith('{'): continue
        d=json.loads(l)
        if d['case']=='networks': print('  net',d['size'],{k:round(v,4) for k,v in d.items() if k.startswith(('R_','G_grad','D_grad','V_grad'))})
        else: print('  step',d['size'],d['attrs'],{k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k in ('grad_cos','grad_l2','grad_relmax','gan_rel','reg_rel','loss_rel','cont_rel')}, 'per_attr', max(d['per_attr_reg_loss_delta']))
PY
tail -3 $OUT/e2
timeout 900 python -m pytest tests/test_trajectory_gpu.py tests/test_kernels_gpu.py -q -m gpu -k "trajectory or bit_stable or bit_identically or winograd or difference_residual" -s > $OUT/pytest_new.log 2>&1
grep -E "trajectory |passed|failed|distinct|Error|assert" $OUT/pytest_new.log | cut -c1-400 | tail -30
