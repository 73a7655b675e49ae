#!/bin/bash
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python $R/bench.py --no-cpu-baseline --steps 24 --sustain-seconds 0 --projection-world 0 --no-online --no-shared-crops > $OUT/prof_bench.log 2>&1
cd $R
python tools/kstats_region.py $OUT/prof $OUT/bench_n1_timed_region_kernel_stats.csv $OUT/bench_n1_isolated_pass_kernel_stats.csv > $OUT/kstats_region.log 2>&1
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/bench_n1_kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/kstats_region.log; head -4 $OUT/bench_n1_isolated_pass_kernel_stats.csv | cut -c1-160
grep '^{' $OUT/prof_bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], r['avg_launch_us'], r['isolated']['avg_launch_us'], r['isolated']['achieved'])"
