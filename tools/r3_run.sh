#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 1800 python -m pytest tests/test_gpu_geometry.py tests/test_gpu_multirank.py tests/test_gpu_pipeline.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | grep -v amdgpu | tail -3
cd /tmp; rm -rf $OUT/prof_e
NOSAM=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_e -- python $R/tools/round_profile.py 8 12 > $OUT/prof_e.log 2>&1
python - <<PY
import csv, glob
f=glob.glob('$OUT/prof_e/**/*kernel_stats.csv', recursive=True)[0]
out=[]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('k_kf_','k_track_project','k_vote_decide','k_backproj_flag')): out.append("%s %.1f" % (r['Name'].split('::')[-1].split('(')[0], float(r['AverageNs'])/1e3))
print(" | ".join(out))
PY
find $OUT -name "*kernel_trace.csv" -delete
cd $R
timeout 300 python tools/replicated_cost.py 64 8 2>&1 | grep -v amdgpu | tail -1
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-online --sustain-seconds 0 --no-roofline --no-shared-crops 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['projection']['ms_per_round'])"; done
