#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_hiera.py -x -q -k "attention or vit_forward_vs_oracle_full_size or hiera_vs_oracle" 2>&1 | tail -2
for i in 1 2; do timeout 600 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline --sustain-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
python tools/enc_table.py vit 4 2>&1 | grep -v amdgpu | head -8
python tools/enc_table.py sam 4 2>&1 | grep -v amdgpu | head -12
