#!/bin/bash
# socket power and shader clock while one fp32 conv shape runs (gpurun)
for shape in "64 64 3 1 1024" "512 512 3 1 64" "256 1024 1 1 64" "64 128 3 2 513" "128 64 3 2 256 T"; do
  echo "== $shape"
  set -- $shape
  python tools/probes/f32_spin.py $1 $2 $3 $4 $5 6 $6 &
  PID=$!
  sleep 3.5
  for i in 1 2; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket" | head -3; sleep 0.5; done
  wait $PID
done
echo "== whole c3 step"
python bench.py --steps 40 --warmup 2 --cpu_baseline_s 0 --no_alt_precision --sweep none --no_kernel_events > /dev/null 2>&1 &
PID=$!
sleep 9
for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket" | head -3; sleep 0.7; done
wait $PID
echo "== whole c5 step"
python bench.py --config c5 --steps 120 --warmup 2 --cpu_baseline_s 0 --no_alt_precision --sweep none --no_kernel_events > /dev/null 2>&1 &
PID=$!
sleep 12
for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Socket" | head -3; sleep 0.7; done
wait $PID
