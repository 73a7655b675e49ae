"""Hiera patch embedding at the bench's 12-frame group: the direct 7 x 7 / stride-4 convolution (`ovo_hiera_patch_embed`) against the
im2col + per-image GEMM form it replaced (`ovo_im2col` + `ovo_gemm` with the position embedding as the GEMM's `add`).
usage: python tools/patch_embed_bench.py [frames] [card]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ovo_amd import _lib as L
from ovo_amd.encoders.hiera import SPECS
B = int(sys.argv[1]) if len(sys.argv) > 1 else 12
spec = SPECS[sys.argv[2] if len(sys.argv) > 2 else "hiera_b+"]
dev = torch.device("cuda", 0)
lib = L.load()
S, E = spec.image_size, spec.embed_dim
T0 = (S // 4) ** 2
img = torch.randn(B, 3, S, S, device=dev)
w = (torch.randn(E, 192, device=dev) * 0.05).to(torch.bfloat16)
w[:, 147:] = 0
bias, pos = torch.randn(E, device=dev), torch.randn(T0, E, device=dev)
out_a, out_b = torch.empty(B, T0, E, device=dev), torch.empty(B, T0, E, device=dev)
col = torch.empty(B * T0, 192, dtype=torch.bfloat16, device=dev)

def direct():
    L.check(lib.ovo_hiera_patch_embed(L.ptr(img), B, S, E, L.ptr(w), 192, L.ptr(bias), L.ptr(pos), L.ptr(out_a), L.stream()))

def two_pass():
    L.check(lib.ovo_im2col(L.ptr(img), B, 3, S, S, 7, 4, 3, L.ptr(col), 192, L.stream()))
    for b in range(B):
        q = L.Gemm()
        q.A, q.lda, q.W, q.ldw, q.bias = col[b * T0:].data_ptr(), 192, w.data_ptr(), 192, bias.data_ptr()
        q.C, q.ldc, q.add, q.ld_add = out_b[b].data_ptr(), E, pos.data_ptr(), E
        q.M, q.N, q.K, q.in_dtype, q.out_dtype, q.act, q.alpha = T0, E, 192, 2, 0, 0, 1.0
        L.check(lib.ovo_gemm(L.C.byref(q), L.stream()))

def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / reps

t_d, t_t = timed(direct), timed(two_pass)
alg = B * (3 * S * S * 4 + T0 * E * 4) + T0 * E * 4
print(f"{spec.name if hasattr(spec, 'name') else ''} B={B} S={S} E={E}: direct {t_d:.1f} us = {alg / t_d / 1e6:.2f} TB/s of algorithmic bytes (image in + tokens out + pos once); "
      f"im2col + {B} GEMMs {t_t:.1f} us; max |difference| {(out_a - out_b).abs().max().item():.2e}")
