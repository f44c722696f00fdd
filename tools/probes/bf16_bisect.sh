run() { echo "== $1"; env $1 python tools/probes/bf16_repeat_g.py 64 4 40 > /tmp/r1.txt 2>&1 & env $1 python tools/probes/bf16_repeat_g.py 64 4 40 > /tmp/r2.txt 2>&1; wait; grep -v amdgpu /tmp/r1.txt | tail -3 | tr '\n' ' '; echo; grep -v amdgpu /tmp/r2.txt | tail -3 | tr '\n' ' '; echo; }
run "X=1"
run "L2I_H8_LEAN=0"
run "L2I_H8_NOSEP=1"
run "L2I_H8_KS=2"
run "L2I_H8_LEAN=0 L2I_H8_NOSEP=1 L2I_H8_KS=2"
