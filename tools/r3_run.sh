#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_encoder.py -x -q -m gpu -k "shared or textregion or lookahead" 2>&1 | tail -8 > gpurun_out/t1.log
timeout 600 python bench.py --no-cpu-baseline --no-online --projection-world 0 --sustain-seconds 0 --no-roofline 2>&1 | grep '^{' > gpurun_out/bench_n1.json
cat gpurun_out/t1.log
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_n1.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("shared_crops"))
PY
