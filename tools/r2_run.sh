#!/bin/bash
cd $GRAFT_REPO_ROOT
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range())"
run() { timeout 300 python bench.py --no-cpu-baseline --no-roofline --sustain-seconds 0 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('$1', b['value'], b['ms_per_step'], b['per_step_ms']['median'], b['per_step_ms']['max'])"; }
run base
OVO_MAIN_PRIORITY=-1 run "main high"
OVO_MAIN_PRIORITY=-1 OVO_SAM_PRIORITY=0 OVO_VIT_PRIORITY=0 run "main high, sides 0"
OVO_SAM_PRIORITY=-1 OVO_VIT_PRIORITY=-1 run "sides high"
run base2
