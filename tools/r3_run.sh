#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_encoder.py -x -q -m gpu -k attention 2>&1 | grep -v amdgpu | tail -2
timeout 600 python tools/attn_bench.py 2>&1 | grep -v amdgpu | cut -c1-110 | head -5
