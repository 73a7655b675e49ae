#!/bin/bash
cd $GRAFT_REPO_ROOT
export TILES="auto,256x128,256x128r" ROUNDS=3
echo "== plain bf16"; SHAPES="4616,4096,1024;4616,3072,1024;16384,1792,448;19600,1344,448;4096,4096,4096;8192,8192,8192" python tools/gemm_bench.py 2>&1 | grep -v amdgpu
echo "== bias gelu"; BIAS=1 ACT=1 SHAPES="4616,4096,1024;16384,1792,448" python tools/gemm_bench.py 2>&1 | grep "^("
echo "== bias add f32"; BIAS=1 ADD=1 INPLACE=1 OUT=f32 SHAPES="4616,1024,4096;4616,1024,1024;16384,448,1792" python tools/gemm_bench.py 2>&1 | grep "^("
