#!/bin/bash
# Evidence run for the packed-fp32-beside-bf16-MFMA finding (gpurun): the same probes on a build whose 16-bit streaming kernels were compiled WITH
# hipcc's SLP vectorizer (tools/ab/libl2i_slp_stream_h8.so, built by hand: hipcc -O3 ... l2i_stream_h8.hip without -fno-slp-vectorize) and on the
# library as shipped (without).
for L in tools/ab/libl2i_slp_stream_h8.so latent2im_amd/libl2i_hip.so; do
  echo "===== library: $L"
  echo "--- one process, conv_h8 launches on stream A, the kernel under test on stream B, 300 repeats on identical inputs (tools/probes/h8_two_streams.py)"
  L2I_ALT_LIB=$L python tools/probes/h8_two_streams.py 300 2>&1 | grep -v amdgpu
  echo "--- two processes at once, each repeating one whole 16-bit training step (64^2, batch 4, three loss-branch streams) 40 times (tools/probes/bf16_repeat.py)"
  L2I_LIB=$L python tools/probes/bf16_repeat.py bf16 64 4 40 > /tmp/r1.txt 2>&1 & L2I_LIB=$L python tools/probes/bf16_repeat.py bf16 64 4 40 > /tmp/r2.txt 2>&1; wait
  grep -v amdgpu /tmp/r1.txt | tail -1 | cut -c1-160; grep -v amdgpu /tmp/r2.txt | tail -1 | cut -c1-160
done
echo "===== standalone instruction probe (tools/probes/pk_beside_mfma.hip): packed vs scalar twins beside a synthetic MFMA spinner — does NOT reproduce the effect"
timeout 300 tools/probes/bin/pk_beside_mfma | cut -c1-330
