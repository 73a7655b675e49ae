#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_e2e.py tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -2
python tools/replicated_cost.py 32 2>&1 | grep -v amdgpu | tail -2
