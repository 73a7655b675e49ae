#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_encoder.py -x -q -k "pingpong" 2>&1 | tail -3
export TILES="auto,256x256,256x128,256x129"
SHAPES="4616,3072,1024;4616,1024,1024;4616,4096,1024;4616,1024,4096;16384,1792,448;16384,448,1792;19600,1344,448;19600,448,448;4096,4096,4096;8192,8192,8192" timeout 900 python tools/gemm_bench.py 2>&1 | grep -v amdgpu
python tools/gemm8p_stamps.py 4616 4096 1024 256x129 2>&1 | grep -v amdgpu
