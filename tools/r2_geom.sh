#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_fullsize.py tests/test_gpu_pipeline.py tests/test_gpu_e2e.py -x -q 2>&1 | tail -3
timeout 600 python tools/geom_bench.py 2>&1 | grep -v amdgpu | tee $OUT/geom_bench.txt
