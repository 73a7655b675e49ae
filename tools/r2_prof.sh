#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
for w in sam vit; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -- python $R/tools/enc_only.py $w 4 10 > $OUT/prof_$w.log 2>&1
  tail -1 $OUT/prof_$w.log
  f=$(find $OUT/prof_$w -name "*kernel_stats.csv" | head -1); cp $f $OUT/prof_${w}_b4_kernel_stats.csv
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per frame:", tot / 1e6 / 13 / 4)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:22]:
    print("%6.3f ms/frame %6d calls %8.1f us avg  %s" % (float(r["TotalDurationNs"]) / 1e6 / 13 / 4, int(r["Calls"]), float(r["AverageNs"]) / 1e3, r["Name"][:110]))
PY
done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
cd $R
timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q 2>&1 | tail -3
