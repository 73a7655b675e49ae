#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_encoder.py -x -q -k "resize" 2>&1 | tail -2
python tools/resize_bench.py 2>&1 | grep -v amdgpu
cp ovo_amd/lib/libovo_hip.so /tmp/new.so; cp ovo_amd/lib/libovo_hip_prev.so ovo_amd/lib/libovo_hip.so; echo "prev lib:"; python tools/resize_bench.py 2>&1 | grep -v amdgpu | grep us; cp /tmp/new.so ovo_amd/lib/libovo_hip.so
