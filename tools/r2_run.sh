#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_hiera.py tests/test_gpu_sam_decoder.py tests/test_gpu_sam_fused.py -x -q 2>&1 | tail -4
export TILES="tiled,stream" ROUNDS=3 ITERS=30
BIAS=1 SHAPES="524288,336,128;262144,336,128" python tools/gemm_bench.py 2>&1 | grep "^("
python tools/enc_only.py sam 8 20 2>&1 | tail -1
OVO_GEMM_NO_STREAM=1 python tools/enc_only.py sam 8 20 2>&1 | tail -1
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-roofline --sustain-seconds 0 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('bench', b['value'], b['ms_per_step'])"; done
OVO_GEMM_NO_STREAM=1 timeout 300 python bench.py --no-cpu-baseline --no-roofline --sustain-seconds 0 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('bench no-stream', b['value'], b['ms_per_step'])"
