#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out
# full default bench (cpu baseline + parity + sustained), then 2 ranks sharing the GPU (gloo) to exercise --gpus 2
timeout 1200 python bench.py > $OUT/r2_bench_full.json 2> $OUT/r2_bench_full.err; tail -3 $OUT/r2_bench_full.err; cat $OUT/r2_bench_full.json
OVO_DIST_BACKEND=gloo OVO_FORCE_DEVICE=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 10 --warmup 4 --no-roofline --sustain-seconds 0 > $OUT/r2_bench_2rank.json 2> $OUT/r2_bench_2rank.err; tail -5 $OUT/r2_bench_2rank.err | cut -c1-300; cat $OUT/r2_bench_2rank.json | cut -c1-1800
