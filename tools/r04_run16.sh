#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r04_run16
cd $R
L2I_DIST_BACKEND=gloo timeout 900 python3 bench.py --gpus 2 --steps 3 --warmup 1 --warmup_s 4 --sweep 4 --config5_steps 3 --event_steps 2 --no_reg_only > gpurun_out/r04_run16/bench_2rank_gloo.json 2> gpurun_out/r04_run16/bench_2rank_gloo.err
echo rc=$?
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_run16/bench_2rank_gloo.json").read().strip().splitlines()[-1])
print(d["n_gpus"], d["value"], d["warmup_steps_run"], d["per_rank_ms_per_step"], d["ranks_seen"], d["config5"]["value"], d["config5"]["warmup_steps_run"])
PY
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_run16/bench.json 2> gpurun_out/r04_run16/bench.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_run16/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["median_ms"], d["min_ms"], d["max_ms"], d["warmup_steps_run"], 'traffic', d["roofline"]["traffic"], 'c5', d["config5"]["value"], d["config5"]["roofline"]["traffic"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["value"], d["cpu_baseline"]["reference_shape"]["value"], d["bench_wall_s"])
PY
