cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_pipeline.py tests/test_gpu_e2e.py tests/test_gpu_multirank.py tests/test_gpu_features.py tests/test_gpu_loopclose.py -x -q > gpurun_out/t1.log 2>&1; tail -8 gpurun_out/t1.log
(echo "running-sum fusion"; python tools/round_emulation.py 8; python tools/round_emulation.py 4; python tools/round_emulation.py 1
python tools/replicated_cost.py 64 8
python tools/round_profile.py 8 24 | head -12) > gpurun_out/round_emulation.txt 2>&1
grep -v amdgpu.ids gpurun_out/round_emulation.txt | cut -c1-150
