#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2; do
python tools/replicated_cost.py 32 2>&1 | grep -v amdgpu | tail -1
OVO_SMALL_TORCH=1 python tools/replicated_cost.py 32 2>&1 | grep -v amdgpu | tail -1
done
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-roofline --sustain-seconds 0 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('bench kernarg', b['value'], b['ms_per_step'], b['per_step_ms']['median'])"
OVO_SMALL_TORCH=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline --sustain-seconds 0 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('bench torch  ', b['value'], b['ms_per_step'], b['per_step_ms']['median'])"
done
