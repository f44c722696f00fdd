#!/bin/bash
# Runs HERE (build container) after `gpurun -- bash tools/r06_collect_and_bench.sh`: copies the summaries that DESIGN.md cites from gpurun_out/ (scratch) into
# profiles/ (tracked).   usage: bash tools/publish_profiles.sh [round tag, default r06]
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
T=${1:-r06}
for CFG in c3 c5; do
  S=$R/gpurun_out/${T}_$CFG
  [ -d "$S" ] || continue
  cp $S/bench_serial_streams.json $R/profiles/${T}_${CFG}_bench_serial_streams.json
  cp $S/stats/run_kernel_stats.csv $R/profiles/${T}_${CFG}_bench_serial_streams_kernel_stats.csv
  cp $S/hbm_traffic.json $R/profiles/${T}_${CFG}_hbm_traffic.json
  cp $S/step_sq_counters_by_kernel.csv $R/profiles/${T}_${CFG}_step_sq_counters_by_kernel.csv
  cp $S/kernel_table.md $R/profiles/${T}_${CFG}_kernel_table.md
  [ -f $R/gpurun_out/${T}_${CFG}_by_shape.txt ] && cp $R/gpurun_out/${T}_${CFG}_by_shape.txt $R/profiles/${T}_${CFG}_by_shape.txt
done
for f in ${T}_bench_c3_driver_cmd_a.json ${T}_bench_c3_driver_cmd_b.json ${T}_bench_c3_driver_cmd_c.json ${T}_bench_c5.json; do
  [ -f $R/gpurun_out/$f ] && cp $R/gpurun_out/$f $R/profiles/$f
done
ls -la $R/profiles | grep ${T}_
