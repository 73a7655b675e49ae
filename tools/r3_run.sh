#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_hiera.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/t1.log
timeout 600 python tools/attn_bench.py > gpurun_out/attn_bench.txt 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-online --projection-world 0 --sustain-seconds 0 2>&1 | grep '^{' > gpurun_out/bench_n1.json
cat gpurun_out/t1.log; grep -v amdgpu gpurun_out/attn_bench.txt | cut -c1-150
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_n1.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("roofline"))
PY
