cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_pipeline.py tests/test_gpu_e2e.py tests/test_gpu_multirank.py -x -q > gpurun_out/t1.log 2>&1; tail -5 gpurun_out/t1.log
(
echo "A chain stream"; python tools/round_emulation.py 8
echo "A0 no chain stream"; OVO_NO_CHAIN_STREAM=1 python tools/round_emulation.py 8
echo "A2 chain stream high priority"; OVO_CHAIN_PRIORITY=-1 python tools/round_emulation.py 8
echo "F world 4"; python tools/round_emulation.py 4
echo "A1 world 1"; python tools/round_emulation.py 1
python tools/round_profile.py 8 24 | head -30) > gpurun_out/round_emulation.txt 2>&1
grep -v amdgpu.ids gpurun_out/round_emulation.txt | cut -c1-150 | head -60
