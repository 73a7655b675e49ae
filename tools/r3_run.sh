#!/bin/bash
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
for v in "OVO_P1_X=2048 OVO_P1_F=512" "OVO_P1_X=1024 OVO_P1_F=512" "OVO_P1_X=512 OVO_P1_F=256" "OVO_P1_X=512 OVO_P1_F=1200" "OVO_P1_X=256 OVO_P1_F=256" "OVO_P1_X=1024 OVO_P1_F=1200"; do
  rm -rf $OUT/prof_e
  env $v NOSAM=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_e -- python $R/tools/round_profile.py 8 12 > $OUT/prof_e.log 2>&1
  python - <<PY
import csv, glob
f=glob.glob('$OUT/prof_e/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'k_kf_phase1' in r['Name']: print("$v: phase1 %.1f us (min %.1f)" % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
done
find $OUT -name "*kernel_trace.csv" -delete
