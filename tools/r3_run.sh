#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
cp ovo_amd/lib/libovo_hip.so /tmp/new.so; cp ovo_amd/lib/libovo_hip_prev.so /tmp/old.so
for v in old new old new; do
  cp /tmp/$v.so ovo_amd/lib/libovo_hip.so
  echo "== $v"
  (ROPE=1 BIAS=1 TILES="auto" SHAPES="13848,3072,1024" python tools/gemm_bench.py | tail -1
   BIAS=1 ADD=1 INPLACE=1 OUT=f32 TILES="auto" SHAPES="13848,1024,1024;13848,1024,4096;49152,448,1792" python tools/gemm_bench.py | tail -3) 2>&1 | grep -v amdgpu
  timeout 300 python tools/enc_only.py vit 12 10 2>&1 | grep -v amdgpu
  timeout 600 python bench.py --no-cpu-baseline --no-online --projection-world 0 --sustain-seconds 0 --no-roofline --no-shared-crops 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'])"
done 2>&1 | tee gpurun_out/ab_epi.txt
cp /tmp/new.so ovo_amd/lib/libovo_hip.so
