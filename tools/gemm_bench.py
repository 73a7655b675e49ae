"""GEMM micro-benchmark over the shapes of the ViT-L/14-336 and hiera_b+ forwards (tile override via OVO_GEMM_TILE)."""
import os; os.environ.setdefault("OVO_KNOBS_DYNAMIC", "1")    # this tool flips OVO_* knobs between launches
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ovo_amd import _lib as L
dev = torch.device("cuda", 0)
lib = L.load()
ROT = int(os.environ.get('ROTATE', '1'))   # > 1: cycle through that many weight buffers (cold weights, like a layer stack)
PAD = int(os.environ.get('PAD', '0'))      # extra elements per row of A and W (row stride K + PAD): breaks power-of-two strides
def run(m, n, k, iters=int(os.environ.get('ITERS', '100'))):
    a = torch.randn(m, k + PAD, device=dev).to(torch.bfloat16); ws = [(torch.randn(n, k + PAD, device=dev) * k ** -0.5).to(torch.bfloat16) for _ in range(ROT)]; w = ws[0]
    odt = {'bf16': (torch.bfloat16, 2), 'f32': (torch.float32, 0)}[os.environ.get('OUT', 'bf16')]
    out = torch.empty(m, n, dtype=odt[0], device=dev)
    g = L.Gemm(); g.A, g.lda, g.W, g.ldw, g.bias, g.C, g.ldc, g.add, g.ld_add = a.data_ptr(), k + PAD, w.data_ptr(), k + PAD, None, out.data_ptr(), n, None, 0
    g.M, g.N, g.K, g.in_dtype, g.out_dtype, g.act, g.alpha = m, n, k, 2, odt[1], int(os.environ.get('ACT', '0')), 1.0
    if os.environ.get('BIAS'):
        bias = torch.randn(n, device=dev) + float(os.environ.get('BIAS_SHIFT', '0')); g.bias = bias.data_ptr()      # BIAS_SHIFT=100: every GELU table lookup hits the last entry (no LDS bank conflicts)
    if os.environ.get('ADD'):
        add = out if os.environ.get('INPLACE') else torch.randn(m, n, device=dev); g.add, g.ld_add = add.data_ptr(), n
    call = lambda: lib.ovo_gemm(C.byref(g), L.stream())
    if os.environ.get('ROPE'):          # PE's QKV projection: rotary embedding of the q and k columns in the epilogue (T = 577, head_dim 64)
        T, hd = 577, 64
        cs, sn = torch.rand(T, hd, device=dev), torch.rand(T, hd, device=dev)
        rope = L.Rope(); rope.cos, rope.sin, rope.T, rope.hd, rope.cols, rope.t0 = cs.data_ptr(), sn.data_ptr(), T, hd, 2 * n // 3, 1
        call = lambda: lib.ovo_gemm_rope(C.byref(g), C.byref(rope), L.stream())
    for _ in range(10): L.check(call())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(iters):
        g.W = ws[i % ROT].data_ptr()
        L.check(call())
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / iters
    return us, 2.0 * m * n * k / us / 1e6
shapes = [(1154, 3072, 1024), (1154, 1024, 1024), (1154, 4096, 1024), (1154, 1024, 4096), (4096, 4096, 4096), (8192, 8192, 8192),
          (65536, 336, 128), (65536, 448, 128), (65536, 112, 448), (16384, 672, 224), (4096, 1344, 448), (4096, 1792, 448), (4096, 448, 1792)]
if os.environ.get('SHAPES'): shapes = [tuple(int(v) for v in t.split(',')) for t in os.environ['SHAPES'].split(';')]
tiles = os.environ.get("TILES", "auto,128x128,64x128,128x64,64x64").split(",")
ROUNDS = int(os.environ.get("ROUNDS", "3"))     # interleaved rounds per shape; the MEDIAN is reported (clock / power state drifts between runs)
print("%-22s" % "M,N,K" + "".join("%22s" % t for t in tiles))
for s in shapes:
    res = {t: [] for t in tiles}
    for _ in range(ROUNDS):
        for t in tiles:
            os.environ.pop("OVO_GEMM_NO_8P", None); os.environ.pop("OVO_GEMM_NO_STREAM", None); os.environ.pop("OVO_8P_TAILWAIT", None)
            os.environ["OVO_8P_MFMA32"] = "0"
            if t.endswith("+w"): os.environ["OVO_8P_TAILWAIT"] = "1"; t0 = t; t = t[:-2]
            elif t.endswith("+32"): os.environ["OVO_8P_MFMA32"] = "1"; t0 = t; t = t[:-3]     # the ping-pong kernel's 32 x 32 x 16 MFMA K-loop
            else: t0 = t
            if t == "auto": os.environ.pop("OVO_GEMM_TILE", None)
            elif t == "tiled": os.environ.pop("OVO_GEMM_TILE", None); os.environ["OVO_GEMM_NO_STREAM"] = "1"   # the tiled kernels' own choice (8p or ring)
            elif t == "ring": os.environ.pop("OVO_GEMM_TILE", None); os.environ["OVO_GEMM_NO_8P"] = "1"; os.environ["OVO_GEMM_NO_STREAM"] = "1"     # the 128-row ring kernels' own choice
            else: os.environ["OVO_GEMM_TILE"] = t
            res[t0].append(run(*s)[0])
    row = "%-22s" % str(s)
    for t in tiles:
        us = sorted(res[t])[len(res[t]) // 2]
        row += "%12.1fus %6.0fTF" % (us, 2.0 * s[0] * s[1] * s[2] / us / 1e6)
    print(row)
