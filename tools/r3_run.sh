#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_fullsize.py tests/test_gpu_multirank.py tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | grep -v amdgpu | tail -3
timeout 300 python tools/geom_bench.py 2>&1 | grep -v amdgpu > gpurun_out/geom_bench.txt; cat gpurun_out/geom_bench.txt
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-online --sustain-seconds 0 --no-roofline --no-shared-crops 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['projection']['ms_per_round'])"; done
