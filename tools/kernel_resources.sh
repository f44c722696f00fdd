#!/bin/bash
# Build container (no GPU): registers, occupancy and SPILLS of every kernel of libl2i_hip.so from the compiler's resource report
# (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel; kernels with spilled registers are marked.  Run before collecting profiles:
# a spilled register is scratch traffic that the PMC passes count as HBM bytes of the launch ("A correction ...").
#   bash tools/kernel_resources.sh [--spills]      (--spills: only the kernels that spill)
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/latent2im_amd/csrc
ONLY=${1:-}
for f in *.hip; do
  for def in "" "-DL2I_H8_F16"; do
    case "$f" in l2i_conv_h8.hip|l2i_stream_h8.hip) ;; *) [ -n "$def" ] && continue ;; esac
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I$R/include -I. -fno-slp-vectorize $def -Rpass-analysis=kernel-resource-usage -c $f -o /dev/null 2>&1 |
      grep -E "Function Name|VGPRs:|AGPRs:|VGPRs Spill|Occupancy|LDS Size" | paste - - - - - - |
      sed "s/$f:[0-9:]* remark: *//g; s/\[-Rpass-analysis=kernel-resource-usage\]//g; s/Function Name: //; s/  */ /g" |
      awk -v only="$ONLY" -v file="$f$def" '{ spill = ($0 !~ /VGPRs Spill: 0 /); if (only != "--spills" || spill) print (spill ? "SPILL " : "      ") file " " $0 }'
  done
done
