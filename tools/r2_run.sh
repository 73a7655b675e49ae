#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/prof_host.py 24 2>&1 | grep -v amdgpu | cut -c1-150 | head -24
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-roofline --sustain-seconds 0 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('bench', b['value'], b['ms_per_step'], b['per_step_ms'])"; done
timeout 1200 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_geometry.py tests/test_gpu_pipeline.py tests/test_gpu_multirank.py -x -q 2>&1 | tail -3
python __graft_entry__.py --smoke 2>&1 | tail -1
