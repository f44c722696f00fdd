#!/bin/bash
# Builds timing-ablation variants of libl2i_hip.so (in the build container) into tools/ab/: the F(4x4,3x3) kernel without its DMA (DMA), its
# transforms (XF), its MFMAs (MFMA), its patch reads (LDSD), its stores (EPI) and combinations.  Results are wrong by construction: timing only
# (tools/probes/w4_ablate_run.sh on the GPU box).   usage: bash tools/probes/w4_ablate.sh "DMA XF MFMA LDSD EPI DMA+XF ..."
set -eu
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/latent2im_amd/csrc
VARIANTS=${1:-"DMA XF MFMA EPI DMA+XF"}
mkdir -p $R/tools/ab
make -C $C -j8 > /dev/null
OBJS=$(cd $C && ls *.o | grep -v l2i_wino4.o | sed "s|^|$C/|")
for v in $VARIANTS; do
  D=""; for x in ${v//+/ }; do if [ $x = UPF ]; then D="$D -DL2I_W4_UPF"; else D="$D -DL2I_W4_ABLATE_$x"; fi; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I$R/include -I$C $D -c $C/l2i_wino4.hip -o /tmp/w4_abl_$v.o &
done
wait
for v in $VARIANTS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/ab/libl2i_w4_$v.so /tmp/w4_abl_$v.o $OBJS
done
ls -la $R/tools/ab/libl2i_w4_*
