OUT=$GRAFT_REPO_ROOT/gpurun_out; R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $R/bench.py --no-cpu-baseline --census $OUT/pmc_census.json --no-shared-crops --steps 24 --warmup 14 --sustain-seconds 0 --projection-world 0 --no-online > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $R/bench.py --no-cpu-baseline --no-roofline --no-shared-crops --steps 24 --warmup 14 --sustain-seconds 0 --projection-world 0 --no-online > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -- python $R/tools/pmc_calib.py > $OUT/calib_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -- python $R/tools/pmc_calib.py > $OUT/calib_write.log 2>&1
cd $R
python tools/pmc_traffic.py --fetch-dir $OUT/pmc_fetch --write-dir $OUT/pmc_write --calib-fetch-dir $OUT/calib_fetch --calib-write-dir $OUT/calib_write --census $OUT/pmc_census.json --out $OUT/pmc_traffic.json > $OUT/pmc_traffic.log 2>&1
tail -8 $OUT/pmc_traffic.log
cd /tmp; timeout 1500 python $R/tools/pmc_shapes.py --out $OUT/pmc_shapes.json > $OUT/pmc_shapes.txt 2>&1; cd $R; tail -25 $OUT/pmc_shapes.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
