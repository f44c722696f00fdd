#!/bin/bash
# On the GPU box: h8_bench on the ablation builds.  usage: bash tools/probes/h8_ablate_run.sh "<variants>" [H8_ONLY]
V=${1:-"MFMA TILE W STORE"}
export H8_ONLY=${2:-k3s1}
for v in base $V; do
    if [ $v = base ]; then L=latent2im_amd/libl2i_hip.so; else L=tools/ab/libl2i_h8_no_$v.so; fi
    echo "== $v"
    python tools/probes/h8_bench.py $L 2>&1 | grep " k[13] s" | head -${NCASE:-5}
  done
