#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_e2e.py tests/test_gpu_loopclose.py -x -q > $OUT/r2_pytest.log 2>&1; tail -5 $OUT/r2_pytest.log
timeout 600 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_hiera.py -x -q -k "golden or flops or remove_global" > $OUT/r2_pytest2.log 2>&1; tail -5 $OUT/r2_pytest2.log
for b in 1 4 8; do timeout 600 python bench.py --encoder-batch $b --no-cpu-baseline --steps 24 --warmup 8 > $OUT/r2_bench_b$b.json 2> $OUT/r2_bench_b$b.err; tail -2 $OUT/r2_bench_b$b.err; cut -c1-2200 $OUT/r2_bench_b$b.json; done
