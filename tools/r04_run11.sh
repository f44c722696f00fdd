#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/r04_run11
cd $R
L2I_DIST_BACKEND=gloo timeout 900 python3 bench.py --gpus 2 --steps 3 --warmup 1 --warmup_s 0 --sweep 4,8 --config5_steps 3 --event_steps 2 > gpurun_out/r04_run11/bench_2rank_gloo.json 2> gpurun_out/r04_run11/bench_2rank_gloo.err
echo rc=$?
tail -c 600 gpurun_out/r04_run11/bench_2rank_gloo.err
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r04_run11/bench_2rank_gloo.json").read().strip().splitlines()[-1])
print(d["n_gpus"], d["value"], d["ms_per_step"], d["per_rank_ms_per_step"], d["ranks_seen"], d["allreduce_us"], d["config5"]["value"], {k:v["value"] for k,v in d["reg_only"].items()}, d["batch_sweep"])
PY
