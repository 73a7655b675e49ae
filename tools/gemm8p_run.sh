#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/gemm8p_stamps.py 4616 3072 1024 256x256 2>&1 | grep -v amdgpu
python tools/gemm8p_stamps.py 4616 3072 1024 256x128 2>&1 | grep -v amdgpu
python tools/gemm8p_stamps.py 4616 1024 4096 256x128 2>&1 | grep -v amdgpu
python tools/gemm8p_stamps.py 4096 4096 4096 256x256 2>&1 | grep -v amdgpu
