#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_encoder.py -x -q -k "attention" 2>&1 | tail -2
python tools/attn_bench.py 2>&1 | grep -v amdgpu
cp ovo_amd/lib/libovo_hip.so /tmp/new.so; cp ovo_amd/lib/libovo_hip_prev.so ovo_amd/lib/libovo_hip.so; echo "prev lib (auto = narrow):"; python tools/attn_bench.py 2>&1 | grep -v amdgpu | cut -c1-60; cp /tmp/new.so ovo_amd/lib/libovo_hip.so
