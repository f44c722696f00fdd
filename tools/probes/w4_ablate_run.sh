#!/bin/bash
# GPU box: times the ablation builds of tools/probes/w4_ablate.sh on three layer shapes (timing only: their results are wrong by construction).
# usage: bash tools/probes/w4_ablate_run.sh [mode=all|r4] > gpurun_out/...txt
R=$(cd "$(dirname "$0")/../.." && pwd)
MODE=${1:-all}
for shape in "512 512 64 8" "64 64 1024 8" "128 128 256 8"; do
  echo "== $shape"
  for kind in plain relu_in; do
    python $R/tools/probes/one_wino4.py $shape $MODE $kind 30
    for lib in $R/tools/ab/libl2i_w4_*.so; do
      L2I_LIB=$lib python $R/tools/probes/one_wino4.py $shape $MODE $kind 30
    done
  done
done
