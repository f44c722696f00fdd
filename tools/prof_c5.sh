#!/bin/bash
# rocprofv3 kernel stats of the c5 (16-bit path) step, serial streams, eager launches (gpurun)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_c5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $R
rocprofv3 --kernel-trace --stats -d $OUT/stats -o run --output-format csv -- python3 bench.py --config c5 --hip_graph 0 --serial_streams --cpu_baseline_s 0 --no_alt_precision --sweep none --steps 3 --warmup 1 --no_kernel_events > $OUT/bench.json 2> $OUT/bench.err
python3 - <<'PY'
import csv, glob, os
out = os.environ.get('GRAFT_REPO_ROOT', os.getcwd()) + '/gpurun_out/prof_c5'
f = glob.glob(out + '/stats/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms over 4 steps: %.1f' % (tot / 1e6))
for r in rows[:40]:
    print('%-90s calls %6s  total %8.2f ms  avg %8.1f us  %5.1f%%' % (r['Name'][:90], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, 100 * float(r['TotalDurationNs']) / tot))
PY
