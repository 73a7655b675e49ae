#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_encoder.py -x -q -m gpu -k "rope or pe or textregion or vit" 2>&1 | grep -v amdgpu | tail -3
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export OVO_ROPE_UNPACKED=1; else unset OVO_ROPE_UNPACKED; fi
  timeout 300 python tools/enc_only.py vit 12 10 2>&1 | grep -v amdgpu | sed "s/^/UNPACKED=$v /"
done
