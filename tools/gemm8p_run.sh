#!/bin/bash
cd $GRAFT_REPO_ROOT
export TILES="ring,128x128,128x64,256x128" SHAPES="262144,448,128;65536,896,256;262144,336,128;65536,224,896"
echo "--- bf16 out"; timeout 600 python tools/gemm_bench.py 2>&1 | grep -v amdgpu
echo "--- f32 out"; OUT=f32 timeout 600 python tools/gemm_bench.py 2>&1 | grep -v amdgpu
