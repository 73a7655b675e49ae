#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_multirank.py tests/test_gpu_e2e.py -x -q > $OUT/r2_pytest.log 2>&1; tail -15 $OUT/r2_pytest.log
for b in 4 8; do timeout 900 python bench.py --encoder-batch $b --steps 24 --warmup 8 --no-cpu-baseline > $OUT/r2_bench_b$b.json 2> $OUT/r2_bench_b$b.err; tail -3 $OUT/r2_bench_b$b.err; cut -c1-3000 $OUT/r2_bench_b$b.json; done
