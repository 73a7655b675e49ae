"""GEMM micro-benchmark over the shapes of the ViT-L/14-336 and hiera_b+ forwards (tile override via OVO_GEMM_TILE)."""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import torch
from ovo_amd import _lib as L
dev = torch.device("cuda", 0)
lib = L.load()
def run(m, n, k, iters=30):
    a = torch.randn(m, k, device=dev).to(torch.bfloat16); w = (torch.randn(n, k, device=dev) * k ** -0.5).to(torch.bfloat16)
    out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    g = L.Gemm(); g.A, g.lda, g.W, g.ldw, g.bias, g.C, g.ldc, g.add, g.ld_add = a.data_ptr(), k, w.data_ptr(), k, None, out.data_ptr(), n, None, 0
    g.M, g.N, g.K, g.in_dtype, g.out_dtype, g.act, g.alpha = m, n, k, 2, 2, 0, 1.0
    for _ in range(3): L.check(lib.ovo_gemm(C.byref(g), L.stream()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): L.check(lib.ovo_gemm(C.byref(g), L.stream()))
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / iters
    return us, 2.0 * m * n * k / us / 1e6
shapes = [(1154, 3072, 1024), (1154, 1024, 1024), (1154, 4096, 1024), (1154, 1024, 4096), (4096, 4096, 4096), (8192, 8192, 8192),
          (65536, 336, 128), (65536, 448, 128), (65536, 112, 448), (16384, 672, 224), (4096, 1344, 448), (4096, 1792, 448), (4096, 448, 1792)]
tiles = os.environ.get("TILES", "auto,128x128,64x128,128x64,64x64").split(",")
print("%-22s" % "M,N,K" + "".join("%22s" % t for t in tiles))
for s in shapes:
    row = "%-22s" % str(s)
    for t in tiles:
        if t == "auto": os.environ.pop("OVO_GEMM_TILE", None)
        else: os.environ["OVO_GEMM_TILE"] = t
        us, tf = run(*s)
        row += "%12.1fus %6.0fTF" % (us, tf)
    print(row)
