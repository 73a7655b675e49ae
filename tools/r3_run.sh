cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_pipeline.py tests/test_gpu_e2e.py tests/test_gpu_multirank.py -x -q > gpurun_out/t1.log 2>&1; tail -5 gpurun_out/t1.log
OVO_ROUND_CHAIN=1 timeout 1200 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_multirank.py -x -q > gpurun_out/t2.log 2>&1; tail -5 gpurun_out/t2.log
