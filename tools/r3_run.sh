#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
OVO_DIST_BACKEND=gloo OVO_FORCE_DEVICE=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 12 --warmup 3 > gpurun_out/bench_2rank.json 2> gpurun_out/bench_2rank.err
echo "rc $?"; grep '^{' gpurun_out/bench_2rank.json | cut -c1-900; tail -3 gpurun_out/bench_2rank.err | cut -c1-300
OVO_DIST_BACKEND=gloo OVO_FORCE_DEVICE=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --steps 6 --warmup 2 --sam-full > gpurun_out/bench_2rank_samfull.json 2> gpurun_out/bench_2rank_samfull.err
echo "rc $?"; grep '^{' gpurun_out/bench_2rank_samfull.json | cut -c1-400; tail -2 gpurun_out/bench_2rank_samfull.err | cut -c1-300
