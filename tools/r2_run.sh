#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_multirank.py -x -q 2>&1 | tail -2
for a in "--steps 20 --warmup 5" "--steps 24 --warmup 3" "--steps 20 --warmup 5"; do timeout 300 python bench.py --gpus 1 $a --no-cpu-baseline --no-roofline --sustain-seconds 0 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('$a', b['value'], b['ms_per_step'])"; done
