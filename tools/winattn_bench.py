"""Hiera stage-1 attention half up to the output projection at the bench's 12-frame group: `ovo_window_attention_f32` (one launch) against the
three launches it replaces, timed through two whole forwards (OVO_HIERA_NO_WINATTN=1 vs default) and alone.
usage: python tools/winattn_bench.py [frames]"""
import os, sys
os.environ.setdefault("OVO_KNOBS_DYNAMIC", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ovo_amd import _lib as L
from ovo_amd.encoders.hiera import SPECS, HipHiera
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda", 0)
lib = L.load()
H = W = 256; C = 112
x = torch.randn(B, H, W, C, device=dev)
g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
w = (torch.randn(3 * C, 128, device=dev) * 0.1).to(torch.bfloat16); w[:, C:] = 0
bias = torch.zeros(3 * C, device=dev)
n_win = B * 32 * 32
att = torch.zeros(n_win * 64, 128, dtype=torch.bfloat16, device=dev)

def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a_.record()
    for _ in range(reps): fn()
    b_.record(); torch.cuda.synchronize()
    return 1e3 * a_.elapsed_time(b_) / reps

def fused():
    L.check(lib.ovo_window_attention_f32(L.ptr(x), B, H, W, 8, C, C, 2, 0, L.ptr(g), L.ptr(b), 1e-6, L.ptr(w), 128, L.ptr(bias), L.ptr(att), 128, L.stream()))
t = timed(fused)
alg = B * H * W * C * (4 + 2)
print(f"B={B}: fused LN + QKV + 8x8 window attention {t:.1f} us = {alg / t / 1e6:.2f} TB/s of algorithmic bytes (x in, attention out)")
w4 = (torch.randn(6 * C, 128, device=dev) * 0.1).to(torch.bfloat16); w4[:, C:] = 0
bias4 = torch.zeros(6 * C, device=dev)
att4 = torch.zeros(n_win * 16, 256, dtype=torch.bfloat16, device=dev)
def fused_pool():
    L.check(lib.ovo_window_attention_f32(L.ptr(x), B, H, W, 8, C, 2 * C, 4, 1, L.ptr(g), L.ptr(b), 1e-6, L.ptr(w4), 128, L.ptr(bias4), L.ptr(att4), 256, L.stream()))
t = timed(fused_pool)
print(f"B={B}: stage-change block (4 heads, q pooled 2 x 2, two passes) {t:.1f} us")
x2 = torch.randn(B, 128, 128, 224, device=dev)
g2, b2 = torch.ones(224, device=dev), torch.zeros(224, device=dev)
w2 = (torch.randn(3 * 224, 256, device=dev) * 0.1).to(torch.bfloat16); w2[:, 224:] = 0
bias2 = torch.zeros(3 * 224, device=dev)
att2 = torch.zeros(B * 128 * 128, 256, dtype=torch.bfloat16, device=dev)
def fused_s2():
    L.check(lib.ovo_window_attention_f32(L.ptr(x2), B, 128, 128, 4, 224, 224, 4, 0, L.ptr(g2), L.ptr(b2), 1e-6, L.ptr(w2), 256, L.ptr(bias2), L.ptr(att2), 256, L.stream()))
t = timed(fused_s2)
print(f"B={B}: stage-2 block (224 channels, 4 heads, 4 x 4 windows, two passes) {t:.1f} us")
enc = HipHiera(SPECS["hiera_b+"], None, dev, 0)
img = torch.randn(B, 3, 1024, 1024, device=dev)
t_on = timed(lambda: enc.forward(img), 5)
os.environ["OVO_HIERA_NO_WINATTN"] = "1"
t_off = timed(lambda: enc.forward(img), 5)
del os.environ["OVO_HIERA_NO_WINATTN"]
print(f"hiera_b+ forward of {B} frames: {t_on / 1e3:.3f} ms with the fused window attention (stages 1-2), {t_off / 1e3:.3f} ms with the three launches ({(t_off - t_on) / 5:.0f} us per fused block, 5 blocks)")
