#!/bin/bash
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
for b in 4 8; do timeout 300 python bench.py --no-cpu-baseline --no-roofline --encoder-batch $b 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('encoder_batch', b['config']['encoder_batch'], b['value'], b['ms_per_step'], b.get('sustained',{}).get('frames_per_s'))"; done
for b in 4 8; do timeout 600 python bench.py --sam-full --no-cpu-baseline --sustain-seconds 0 --encoder-batch $b 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('sam-full', b['config']['encoder_batch'], b['value'], b['ms_per_step'])"; done
