#!/bin/bash
# Builds timing-ablation variants of libl2i_hip.so (HERE, in the build container) into tools/ab/: the pair kernel without its first / second conv's MFMAs, without the
# operand DMA, the weight DMA, the stores of the wide map, the per-chunk barrier.  Results are wrong by construction: timing only (tools/probes/pair_bench.py with L2I_LIB).
# usage: bash tools/probes/pair_ablate.sh "MFMA1 MFMA2 MFMA1+MFMA2 RES STORE RES+STORE W BAR"
set -eu
R=$(cd "$(dirname "$0")/../.." && pwd)
C=$R/latent2im_amd/csrc
VARIANTS=${1:-"MFMA1 MFMA2 MFMA1+MFMA2 RES STORE RES+STORE W BAR"}
mkdir -p $R/tools/ab /tmp/pair_abl_objs
F="-O3 -std=c++17 -fPIC -fno-slp-vectorize -I$R/include -I$C"
(cd $C && make -j8 >/dev/null)
for v in $VARIANTS; do
  D=""; for x in ${v//+/ }; do D="$D -DL2I_PAIR_ABL_$x"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $F $D -c $C/l2i_pair_h8.hip -o /tmp/pair_abl_$v.o &
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $F $D -DL2I_H8_F16 -c $C/l2i_pair_h8.hip -o /tmp/pair_abl_${v}_f16.o &
done
wait
OTHERS=$(cd $C && ls *.o | grep -v l2i_pair_h8 | sed "s|^|$C/|")
for v in $VARIANTS; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/ab/libl2i_pair_no_$v.so /tmp/pair_abl_$v.o /tmp/pair_abl_${v}_f16.o $OTHERS
done
ls -la $R/tools/ab/libl2i_pair_no_*
