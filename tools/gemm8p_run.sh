#!/bin/bash
cd $GRAFT_REPO_ROOT
OVO_8P_PF=1 timeout 600 python -m pytest tests/test_gpu_encoder.py -x -q -k "pingpong" 2>&1 | tail -2
export TILES="256x256,256x128" SHAPES="4616,3072,1024;4616,4096,1024;4616,1024,4096;16384,1792,448;4096,4096,4096;8192,8192,8192"
echo "--- no prefetch"; timeout 600 python tools/gemm_bench.py 2>&1 | grep -v amdgpu
echo "--- L2 prefetch"; OVO_8P_PF=1 timeout 600 python tools/gemm_bench.py 2>&1 | grep -v amdgpu
OVO_8P_PF=1 python tools/gemm8p_stamps.py 4616 3072 1024 256x256 2>&1 | grep -v amdgpu | grep "k-loop  "
