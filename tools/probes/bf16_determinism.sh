#!/bin/bash
# Two independent single-process runs of the dp_worker step on the 16-bit path at the same time on one GPU (what two ranks do to each other in
# the one-GPU data-parallel tests), repeated: identical inputs must give identical losses every time.  usage: bash tools/probes/bf16_determinism.sh [precision] [rounds]
P=${1:-bf16}
N=${2:-5}
mkdir -p /tmp/det
for i in $(seq 1 $N); do
  L2I_PRECISION=$P python tests/dp_worker.py /tmp/det/a$i.npz 64 4 2 0 > /dev/null 2>&1 &
  PA=$!
  L2I_PRECISION=$P python tests/dp_worker.py /tmp/det/b$i.npz 64 4 2 0 > /dev/null 2>&1 &
  PB=$!
  wait $PA; wait $PB
done
python3 - <<PY
import numpy as np, glob
ref = None
for f in sorted(glob.glob('/tmp/det/*.npz')):
    d = np.load(f)
    l = d['losses'].reshape(-1)
    g = d['grads'].reshape(-1)
    if ref is None: ref = (l, g)
    cos = float((g * ref[1]).sum() / (np.linalg.norm(g) * np.linalg.norm(ref[1])))
    print(f.split('/')[-1], ' '.join('%.8f' % v for v in l), ' grad cos vs first %.6f' % cos, ' max|dl| %.2e' % np.abs(l - ref[0]).max())
PY
