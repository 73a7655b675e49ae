#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_geometry.py -x -q -m gpu 2>&1 | grep -v amdgpu | tail -2
ITERS=50 timeout 300 python tools/geom_bench.py 2>&1 | grep "track_project"
