#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
python tools/enc_table.py sam 12 2>&1 | grep -v amdgpu > gpurun_out/enc_sam.txt
timeout 600 python bench.py --no-cpu-baseline --no-online --projection-world 0 --sustain-seconds 0 --no-roofline --no-shared-crops 2>&1 | grep '^{' > gpurun_out/bench_n1.json
timeout 900 python -m pytest tests/test_gpu_hiera.py tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -3
head -30 gpurun_out/enc_sam.txt
python - <<PY
import json
d=json.loads(open("gpurun_out/bench_n1.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
PY
