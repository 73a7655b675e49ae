cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_features.py -x -q -k "preprocess or resize or crop" > gpurun_out/t1.log 2>&1; tail -15 gpurun_out/t1.log
