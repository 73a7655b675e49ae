#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
timeout 1800 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_multirank.py tests/test_gpu_pipeline.py tests/test_gpu_sam_decoder.py -x -q -m gpu 2>&1 | grep -v amdgpu | tail -3
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_chain2 -- python $R/tools/replicated_cost.py 64 8 > $OUT/prof_chain.log 2>&1
cd $R
python - <<PY
import csv, glob
f=glob.glob('$OUT/prof_chain2/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('fuse_publish','track_project','vote_decide','kf_finish')): print(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3)
PY
tail -2 $OUT/prof_chain.log
find $OUT -name "*kernel_trace.csv" -delete
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-online --sustain-seconds 0 --no-roofline --no-shared-crops 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['projection']['ms_per_round'])"; done
