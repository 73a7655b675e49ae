#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_sam_decoder.py -x -q 2>&1 | tail -15
timeout 300 python tools/amg_bench.py 16 2>&1 | grep -v amdgpu
