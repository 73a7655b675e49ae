#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python bench.py --map-points 5000000 --no-cpu-baseline --no-online --no-shared-crops --sustain-seconds 0 > gpurun_out/bench_5m.json 2> gpurun_out/bench_5m.err; echo "rc $?"
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_5m.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["map_points"], d["projection"]["ms_per_round"], d["projection"]["frames_per_s_if_hidden_exchange"])
PY
tail -2 gpurun_out/bench_5m.err
