#!/bin/bash
# scratch: tile-order sweep of the ping-pong GEMM (OVO_GEMM_STRIP = n-tiles per column strip; 0 = row-major chunks)
cd $GRAFT_REPO_ROOT
OVO_GEMM_STRIP=5 timeout 600 python -m pytest tests/test_gpu_encoder.py -x -q -k "pingpong" 2>&1 | tail -2
export TILES="256x256,256x128" SHAPES="4616,3072,1024;4616,4096,1024;4616,1024,4096;16384,1792,448;19600,1344,448;4096,4096,4096;8192,8192,8192"
for w in 0 2 4 6 8 12; do echo "--- strip $w"; OVO_GEMM_STRIP=$w timeout 600 python tools/gemm_bench.py 2>&1 | grep -v amdgpu; done
