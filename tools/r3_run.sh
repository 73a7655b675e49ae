cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_pipeline.py tests/test_gpu_e2e.py tests/test_gpu_multirank.py -x -q > gpurun_out/t1.log 2>&1; tail -15 gpurun_out/t1.log
(timeout 300 python tools/replicated_cost.py 32 8; HOST=1 timeout 300 python tools/replicated_cost.py 32 1; timeout 300 python tools/replicated_cost.py 32 1; PROFILE=1 timeout 300 python tools/replicated_cost.py 32 8) > gpurun_out/replicated_cost.txt 2>&1; head -60 gpurun_out/replicated_cost.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err; cut -c1-600 gpurun_out/bench_a.json; tail -3 gpurun_out/bench_a.err
