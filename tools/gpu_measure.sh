#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, kernel-trace stats and the PMC traffic passes.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
timeout 300 python tools/query_bench.py > $OUT/query_bench.log 2>&1
timeout 300 python tools/amg_bench.py 16 > $OUT/amg_bench.log 2>&1
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout 600 python bench.py --sam-full --no-cpu-baseline > $OUT/bench_sam_full.json 2> $OUT/bench_sam_full.err
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python $R/bench.py --no-cpu-baseline --steps 10 > $OUT/prof_bench.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_sam_full -- python $R/bench.py --no-cpu-baseline --no-roofline --sam-full --steps 10 > $OUT/prof_sam_full.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $R/bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 2 > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $R/bench.py --no-cpu-baseline --no-roofline --steps 3 --warmup 2 > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -- python $R/tools/pmc_calib.py > $OUT/calib_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -- python $R/tools/pmc_calib.py > $OUT/calib_write.log 2>&1
cd $R
python tools/pmc_traffic.py --fetch-dir $OUT/pmc_fetch --write-dir $OUT/pmc_write --calib-fetch-dir $OUT/calib_fetch --calib-write-dir $OUT/calib_write --out $OUT/pmc_traffic.json > $OUT/pmc_traffic.log 2>&1
# keep the merge-back small: drop the per-dispatch traces, keep stats + reduced PMC
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +8M -delete; find $OUT -name "*agent_info.csv" -delete
tail -3 $OUT/pytest_gpu.log; tail -2 $OUT/smoke.log; cat $OUT/query_bench.log; head -4 $OUT/amg_bench.log; cat $OUT/bench.json; cut -c1-260 $OUT/bench_sam_full.json; tail -16 $OUT/pmc_traffic.log
