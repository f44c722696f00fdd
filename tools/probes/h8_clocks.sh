for shape in "64 64 3 1 1024" "512 512 3 1 64" "64 256 1 1 256"; do
  echo "== $shape"
  python tools/probes/h8_spin.py $shape 6 &
  PID=$!
  sleep 3.5
  for i in 1 2 3; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Socket" | head -6; sleep 0.5; done
  wait $PID
done
echo "== idle"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|Socket" | head -4
