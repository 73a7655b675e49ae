#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_multirank.py tests/test_gpu_sam_decoder.py tests/test_gpu_hiera.py tests/test_gpu_encoder.py -x -q 2>&1 | tail -3
for a in "" "--sam-full"; do timeout 600 python bench.py $a --steps 40 --warmup 8 --no-cpu-baseline --no-roofline --sustain-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['sam2'])"; done
