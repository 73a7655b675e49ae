cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_hiera.py tests/test_gpu_sam_decoder.py -x -q -s -k "oracle" > gpurun_out/t1.log 2>&1; grep -v amdgpu gpurun_out/t1.log | grep -v "^$" | tail -40
