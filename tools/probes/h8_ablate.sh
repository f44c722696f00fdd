#!/bin/bash
# Builds timing-ablation variants of libl2i_hip.so (HERE, in the build container) into tools/ab/: the h8 conv kernel without its MFMAs, without the
# tile DMA, without the weight DMA, without the epilogue's store, and combinations.  Results are wrong by construction: timing only
# (tools/probes/h8_bench.py <variant.so>, tools/probes/h8_ablate_run.sh).   usage: bash tools/probes/h8_ablate.sh "MFMA TILE W STORE TILE+W TILE+W+MFMA ..."
set -eu
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/latent2im_amd/csrc
VARIANTS=${1:-"MFMA TILE W STORE TILE+W TILE+W+MFMA TILE+W+STORE"}
mkdir -p $R/tools/ab /tmp/h8_abl_objs
F="-O3 -std=c++17 -fPIC -fno-slp-vectorize -I$R/include -I$C"
for f in $(cd $C && ls *.hip | grep -v l2i_conv_h8.hip); do
  [ /tmp/h8_abl_objs/${f%.hip}.o -nt $C/$f ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 $F -c $C/$f -o /tmp/h8_abl_objs/${f%.hip}.o &
done
# [r6] the library also carries the fp16 twins of the two h8 files (entry points l2i_*_h8_f16): the ablation switches apply to both conv_h8 objects
[ /tmp/h8_abl_objs/l2i_stream_h8_f16.o -nt $C/l2i_stream_h8.hip ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 $F -DL2I_H8_F16 -c $C/l2i_stream_h8.hip -o /tmp/h8_abl_objs/l2i_stream_h8_f16.o &
for v in $VARIANTS; do
  D=""; for x in ${v//+/ }; do D="$D -DL2I_H8_ABLATE_$x"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $F $D -c $C/l2i_conv_h8.hip -o /tmp/h8_abl_$v.o &
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $F $D -DL2I_H8_F16 -c $C/l2i_conv_h8.hip -o /tmp/h8_abl_${v}_f16.o &
done
wait
for v in $VARIANTS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/ab/libl2i_h8_no_$v.so /tmp/h8_abl_$v.o /tmp/h8_abl_${v}_f16.o /tmp/h8_abl_objs/*.o
done
ls -la $R/tools/ab/libl2i_h8_no_*
