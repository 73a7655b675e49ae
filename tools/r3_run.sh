#!/bin/bash
# scratch driver of one gpurun call (round 3): encoder batch sweep
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
{
for B in 10 12 14 16; do timeout 300 python tools/enc_only.py vit $B 10; done
for B in 8 12 14 16; do timeout 300 python tools/enc_only.py sam $B 6; done
} > gpurun_out/enc_sweep.txt 2>&1
for B in 12 14 16; do
  timeout 600 python bench.py --encoder-batch $B --steps $((4*B)) --warmup $B --no-cpu-baseline --no-roofline --no-online --projection-world 0 --sustain-seconds 0 2>&1 | grep '^{' > gpurun_out/bench_b$B.json
done
tail -n 20 gpurun_out/enc_sweep.txt
for B in 12 14 16; do python - <<PY
import json
d=json.loads(open("gpurun_out/bench_b$B.json").read().strip().splitlines()[-1])
print($B, d["value"], d["ms_per_step"])
PY
done
