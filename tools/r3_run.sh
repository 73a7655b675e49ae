#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1800 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_hiera.py tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | grep -v amdgpu | tail -3
cp ovo_amd/lib/libovo_hip.so /tmp/new.so; cp ovo_amd/lib/libovo_hip_prev.so /tmp/old.so
python - <<'PY'
# bit-equality of the two libraries' LayerNorm on random rows (k_layernorm through the C ABI)
import ctypes as C, torch
outs = []
for lib_path in ("/tmp/old.so", "/tmp/new.so"):
    lib = C.CDLL(lib_path)
    lib.ovo_layernorm.restype = C.c_int
    lib.ovo_layernorm.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
    res = []
    for rows, d in ((1154, 1024), (333, 448), (77, 1152), (500, 112), (129, 768)):
        g = torch.Generator().manual_seed(rows + d)
        x = (torch.randn(rows, d, generator=g) * 3 + 1).cuda(); gm = torch.randn(d, generator=g).cuda(); bt = torch.randn(d, generator=g).cuda()
        for dt, code in ((torch.float32, 0), (torch.bfloat16, 2)):
            y = torch.empty(rows, d, dtype=dt, device="cuda")
            assert lib.ovo_layernorm(x.data_ptr(), d, rows, d, gm.data_ptr(), bt.data_ptr(), 1e-5, y.data_ptr(), d, code, None) == 0
            torch.cuda.synchronize(); res.append(y.clone())
    outs.append(res)
print("k_layernorm old == new:", all(torch.equal(a, b) for a, b in zip(*outs)))
PY
for v in old new old new; do
  cp /tmp/$v.so ovo_amd/lib/libovo_hip.so
  echo "== $v $(timeout 300 python tools/enc_only.py vit 12 10 2>&1 | grep -v amdgpu) | $(timeout 300 python tools/enc_only.py sam 12 6 2>&1 | grep -v amdgpu)"
  timeout 600 python bench.py --no-cpu-baseline --no-online --projection-world 0 --sustain-seconds 0 --no-roofline --no-shared-crops 2>&1 | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'])"
done
cp /tmp/new.so ovo_amd/lib/libovo_hip.so
