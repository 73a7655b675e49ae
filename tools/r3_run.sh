#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_multirank.py tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | grep -v amdgpu | tail -4 > gpurun_out/t1.log
(for w in 1 8; do echo "world $w"; timeout 300 python tools/round_emulation.py $w; done) 2>&1 | grep -v amdgpu | cut -c1-110 > gpurun_out/round_emulation.txt
timeout 300 python tools/replicated_cost.py 64 8 2>&1 | grep -v amdgpu | tail -3 >> gpurun_out/round_emulation.txt
cat gpurun_out/t1.log gpurun_out/round_emulation.txt
