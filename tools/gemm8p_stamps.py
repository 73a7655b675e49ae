"""(needs a debug build: python -m ovo_amd.build --force --gemm-debug)
Timeline of one launch of the 256-row ping-pong GEMM: s_memrealtime (100 MHz) per workgroup at start / K-loop entry / epilogue
entry / end (OVO_8P_STAMPS).  usage: python tools/gemm8p_stamps.py M N K [tile]"""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
from ovo_amd import _lib as L
m, n, k = (int(v) for v in sys.argv[1:4])
tile = sys.argv[4] if len(sys.argv) > 4 else "256x256"
os.environ["OVO_GEMM_TILE"] = tile
bm, bn = (int(v) for v in tile.split("x"))
dev = torch.device("cuda", 0)
lib = L.load()
a = torch.randn(m, k, device=dev).to(torch.bfloat16); w = (torch.randn(n, k, device=dev) * k ** -0.5).to(torch.bfloat16)
f32 = os.environ.get('OUT') == 'f32'
out = torch.empty(m, n, dtype=torch.float32 if f32 else torch.bfloat16, device=dev)
bias = torch.randn(n, device=dev)
tiles = -(-m // bm) * -(-n // bn)
st = torch.zeros(tiles + 8, 4, dtype=torch.int64, device=dev)
g = L.Gemm(); g.A, g.lda, g.W, g.ldw, g.bias, g.C, g.ldc, g.add, g.ld_add = a.data_ptr(), k, w.data_ptr(), k, None, out.data_ptr(), n, None, 0
g.M, g.N, g.K, g.in_dtype, g.out_dtype, g.act, g.alpha = m, n, k, 2, 0 if f32 else 2, int(os.environ.get('ACT', '0')), 1.0
if os.environ.get('BIAS'): g.bias = bias.data_ptr()
if os.environ.get('ADD'): g.add, g.ld_add = out.data_ptr(), n      # in-place residual
for _ in range(5): L.check(lib.ovo_gemm(C.byref(g), L.stream()))
torch.cuda.synchronize()
os.environ["OVO_8P_STAMPS"] = hex(st.data_ptr())
L.check(lib.ovo_gemm(C.byref(g), L.stream()))
torch.cuda.synchronize()
s = st[:tiles].cpu().double() * 0.01          # us
t0 = s[:, 0].min()
s = s - t0
import numpy as np
s = s.numpy()
print(f"{tile} ({m},{n},{k}): {tiles} workgroups")
for name, col in (("start", 0), ("k-loop entry", 1), ("epilogue entry", 2), ("end", 3)):
    print(f"  {name:15s} min {s[:, col].min():7.2f}  median {np.median(s[:, col]):7.2f}  max {s[:, col].max():7.2f} us")
d = np.diff(s, axis=1)
for name, col in (("prologue", 0), ("k-loop", 1), ("epilogue", 2)):
    print(f"  {name:15s} per-workgroup duration: min {d[:, col].min():6.2f} median {np.median(d[:, col]):6.2f} max {d[:, col].max():6.2f} us")
