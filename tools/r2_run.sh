#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_sam_decoder.py -x -q -s -k "vs_oracle" 2>&1 | grep "mask logits\|passed\|failed"
OVO_SAM_RES16=1 timeout 900 python -m pytest tests/test_gpu_sam_decoder.py -x -q -s -k "vs_oracle" 2>&1 | grep "mask logits\|passed\|failed"
OVO_SAM_RES16=1 timeout 300 python tools/amg_bench.py 16 2>&1 | grep "decoder"
