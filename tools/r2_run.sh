#!/bin/bash
cd $GRAFT_REPO_ROOT
for b in 3 4 5 6 7 8; do echo "B=$b"; timeout 600 python bench.py --encoder-batch $b --steps 42 --warmup 7 --no-cpu-baseline --no-roofline --sustain-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
