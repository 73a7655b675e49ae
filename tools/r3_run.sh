cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_fullsize.py tests/test_gpu_e2e.py tests/test_gpu_pipeline.py -x -q > gpurun_out/t1.log 2>&1; tail -5 gpurun_out/t1.log
timeout 600 python tools/geom_bench.py > gpurun_out/geom_bench.txt 2>&1; grep -v amdgpu gpurun_out/geom_bench.txt
