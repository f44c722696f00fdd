#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r04_run17
mkdir -p $OUT
cd $R
for epi in b br rmo; do for lib in "" tools/ab/libl2i_h8_noslot.so; do
  echo "== H8_EPI=$epi lib=${lib:-default}"; H8_EPI=$epi H8_ONLY=k1s1 python3 tools/probes/h8_bench.py $lib 2>&1 | grep -v amdgpu; H8_EPI=$epi H8_ONLY=k3s1 python3 tools/probes/h8_bench.py $lib 2>&1 | grep -v amdgpu | head -4
done; done > $OUT/h8_ab.txt 2>&1
cat $OUT/h8_ab.txt | cut -c1-200
LEAN="--config c5 --steps 20 --warmup 5 --cpu_baseline_s 0 --no_reg_only --sweep none --no_kernel_events"
for i in 1 2; do
  python3 bench.py $LEAN > $OUT/c5_default_$i.json 2>/dev/null
  L2I_LIB=$R/tools/ab/libl2i_h8_noslot.so python3 bench.py $LEAN > $OUT/c5_noslot_$i.json 2>/dev/null
done
python3 - <<'PY'
import json,glob,os
R=os.environ.get('GRAFT_REPO_ROOT','.')
for f in sorted(glob.glob(R+'/gpurun_out/r04_run17/c5_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(os.path.basename(f), d['value'], d['median_ms'], d['min_ms'])
PY
