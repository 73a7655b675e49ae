#!/bin/bash
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python $R/bench.py --no-cpu-baseline --steps 24 --sustain-seconds 0 --projection-world 0 --no-online --no-shared-crops > $OUT/prof_bench.log 2>&1
cd $R
python tools/kstats_region.py $OUT/prof $OUT/bench_n1_timed_region_kernel_stats.csv > $OUT/kstats_region.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete
cat $OUT/kstats_region.log; grep "resize\|im2col\|depth_filter\|kf_phase1" $OUT/bench_n1_timed_region_kernel_stats.csv | cut -c1-160
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline --no-online --projection-world 0 --sustain-seconds 0 --no-roofline --no-shared-crops 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; done
