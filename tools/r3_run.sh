#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_hiera.py -x -q -m gpu -s -k "operand_load" 2>&1 | grep -v amdgpu | grep "fused\|passed\|failed\|Error\|assert" | tail -30 > gpurun_out/t1.log
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -v amdgpu | tail -4 >> gpurun_out/t1.log
cat gpurun_out/t1.log
