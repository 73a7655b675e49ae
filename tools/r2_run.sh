#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_sam_decoder.py tests/test_gpu_sam_fused.py -x -q 2>&1 | tail -4
timeout 300 python tools/amg_bench.py 16 2>&1 | grep "decoder"
