#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_geometry.py -x -q -m gpu -k "many_masks or queued" 2>&1 | grep -v amdgpu | tail -15
