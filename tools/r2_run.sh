#!/bin/bash
cd $GRAFT_REPO_ROOT
one() {
  echo "== $1"
  python tools/enc_only.py sam 8 20 2>&1 | tail -1
  timeout 300 python tools/amg_bench.py 16 2>&1 | grep "decoder"
  timeout 300 python bench.py --no-cpu-baseline --no-roofline --sustain-seconds 0 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('bench', b['value'], b['ms_per_step'])"
}
one "nontemporal stream/skinny stores"
cp ovo_amd/lib/libovo_hip.so /tmp/new.so; cp ovo_amd/lib/libovo_hip_prev.so ovo_amd/lib/libovo_hip.so
one prev
cp /tmp/new.so ovo_amd/lib/libovo_hip.so
timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_sam_fused.py -x -q 2>&1 | tail -2
