"""Hiera stage 3's attention output projection (58800 x 448 x 448, window order -> spatial rows, + residual) followed by norm2: the full-row tile with the
LayerNorm in its epilogue (`ovo_gemm_rowln`) against `ovo_gemm_unwindow` + `ovo_gemm_f32a`'s LayerNorm pass (k_ln_window).   python tools/rowln_bench.py"""
import ctypes as C, os, sys
os.environ.setdefault("OVO_KNOBS_DYNAMIC", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ovo_amd import _lib as L
from ovo_amd.encoders.hiera import SPECS, HipHiera
dev = torch.device("cuda", 0)
lib = L.load()
B, H, ws, N, K = 12, 64, 14, 448, 448
nw = -(-H // ws); M = B * nw * nw * ws * ws
A = torch.randn(M, K, device=dev).to(torch.bfloat16); W = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
bias, x = torch.zeros(N, device=dev), torch.randn(B * H * H, N, device=dev)
g, b = torch.ones(N, device=dev), torch.zeros(N, device=dev)
h = torch.empty(B * H * H, N, dtype=torch.bfloat16, device=dev)
q = L.Gemm()
q.A, q.lda, q.W, q.ldw, q.bias, q.C, q.ldc, q.add, q.ld_add = A.data_ptr(), K, W.data_ptr(), K, bias.data_ptr(), x.data_ptr(), N, x.data_ptr(), N
q.M, q.N, q.K, q.in_dtype, q.out_dtype, q.act, q.alpha = M, N, K, 2, 0, 0, 1.0
win = L.Window(B, H, H, ws, ws)
def timed(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a_.record()
    for _ in range(reps): fn()
    b_.record(); torch.cuda.synchronize()
    return 1e3 * a_.elapsed_time(b_) / reps
t_f = timed(lambda: L.check(lib.ovo_gemm_rowln(C.byref(q), C.byref(win), L.ptr(g), L.ptr(b), 1e-6, L.ptr(h), N, L.stream())))
t_p = timed(lambda: L.check(lib.ovo_gemm_unwindow(C.byref(q), C.byref(win), L.stream())))
t_l = timed(lambda: L.check(lib.ovo_layernorm(L.ptr(x), N, B * H * H, N, L.ptr(g), L.ptr(b), 1e-6, L.ptr(h), N, 2, L.stream())))
print(f"projection + residual + norm2, full-row tile: {t_f:.1f} us; projection + residual alone (128 x 64 tiles): {t_p:.1f} us; LayerNorm pass: {t_l:.1f} us")
enc = HipHiera(SPECS["hiera_b+"], None, dev, 0)
img = torch.randn(B, 3, 1024, 1024, device=dev)
os.environ["OVO_HIERA_PROJ_LN"] = "1"
t_on = timed(lambda: enc.forward(img), 5)
del os.environ["OVO_HIERA_PROJ_LN"]
t_off = timed(lambda: enc.forward(img), 5)
print(f"hiera_b+ forward of {B} frames: {t_on / 1e3:.3f} ms with it, {t_off / 1e3:.3f} ms without ({(t_off - t_on) / 16:.1f} us per stage-3 block)")
