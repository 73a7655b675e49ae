#!/bin/bash
cd $GRAFT_REPO_ROOT
export TILES="auto" ROUNDS=3
for d in 0 4 8 12 16 24; do echo "== delay $d us"; OVO_8P_DELAY=$d BIAS=1 ACT=1 SHAPES="9232,4096,1024;32768,1792,448;9232,3072,1024;16384,4096,1024;8192,8192,8192" python tools/gemm_bench.py 2>&1 | grep "^("; done
