#!/bin/bash
# Runs HERE (build container) after `gpurun -- bash tools/collect_profiles.sh c3|c5` and the two plain bench runs: copies the summaries that
# DESIGN.md cites from gpurun_out/ (scratch) into profiles/ (tracked).
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
for CFG in c3 c5; do
  S=$R/gpurun_out/r05_$CFG
  [ -d "$S" ] || continue
  cp $S/bench_serial_streams.json $R/profiles/r05_${CFG}_bench_serial_streams.json
  cp $S/stats/run_kernel_stats.csv $R/profiles/r05_${CFG}_bench_serial_streams_kernel_stats.csv
  cp $S/hbm_traffic.json $R/profiles/r05_${CFG}_hbm_traffic.json
  cp $S/step_sq_counters_by_kernel.csv $R/profiles/r05_${CFG}_step_sq_counters_by_kernel.csv
  cp $S/kernel_table.md $R/profiles/r05_${CFG}_kernel_table.md
done
for f in r05_bench_c3.json r05_bench_c5.json r05_bench_c5_eager_launches.json; do
  [ -f $R/gpurun_out/$f ] && cp $R/gpurun_out/$f $R/profiles/$f
done
ls -la $R/profiles | grep r05_
