#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_hiera.py tests/test_gpu_sam_decoder.py tests/test_gpu_pipeline.py -x -q -m gpu -s 2>&1 | grep -v amdgpu | grep "fused vs\|passed\|failed\|Error" | tail -10
for v in 0 1 0 1; do
  if [ $v = 1 ]; then export OVO_NO_LN_FOLD=1; else unset OVO_NO_LN_FOLD; fi
  timeout 600 python bench.py --no-cpu-baseline --no-online --projection-world 0 --sustain-seconds 0 --no-roofline --no-shared-crops 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NO_LN_FOLD=$v', d['value'], d['ms_per_step'])"
done | tee gpurun_out/ab.txt
