"""Attention micro-benchmark over the ViT-L/14-336 and hiera_b+ shapes (12 frames per launch, as the bench's look-ahead groups):
auto = what ovo_attention picks; a32 = k_attention32 forced (OVO_ATTN32=1); narrow / wide = the 16 x 16-tile kernel with 64 / 128-query
workgroups (OVO_ATTN32=0); notiny = OVO_ATTN_NO_TINY on top (the tiled kernel for the <= 64-token problems).  GB/s = the q / k / v / o bytes of the launch."""
import os; os.environ.setdefault("OVO_KNOBS_DYNAMIC", "1")    # this tool flips OVO_* knobs between launches
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ovo_amd import _lib as L
dev = torch.device("cuda", 0); lib = L.load()
def run(B, H, Tq, Tk, hd, iters=30):
    D = H * hd; T = max(Tq, Tk)
    qkv = torch.randn(B, T, 3, H, hd, device=dev).to(torch.bfloat16)
    out = torch.zeros(B, Tq, D, dtype=torch.bfloat16, device=dev)
    a = L.Attention(); base = qkv.data_ptr()
    a.q, a.k, a.v, a.o = base, base + D * 2, base + 4 * D, out.data_ptr()
    a.q_sb = a.k_sb = a.v_sb = T * 3 * D; a.q_sh = a.k_sh = a.v_sh = hd; a.q_st = a.k_st = a.v_st = 3 * D
    a.o_sb, a.o_sh, a.o_st = Tq * D, hd, D
    a.B, a.H, a.Tq, a.Tk, a.hd, a.scale = B, H, Tq, Tk, hd, hd ** -0.5
    for _ in range(3): L.check(lib.ovo_attention(C.byref(a), L.stream()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): L.check(lib.ovo_attention(C.byref(a), L.stream()))
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / iters
    return us, 4.0 * B * H * Tq * Tk * hd / us / 1e6, 2.0 * B * H * hd * (2 * Tq + 2 * Tk) / us / 1e3
shapes = [(24, 16, 577, 577, 64), (2, 16, 577, 577, 64), (12, 8, 4096, 4096, 56), (300, 8, 196, 196, 56), (12288, 2, 64, 64, 56), (12288, 4, 16, 64, 56), (12288, 4, 16, 16, 56),
          (12288, 8, 4, 16, 56), (300, 16, 49, 196, 56), (300, 16, 49, 49, 56), (8, 16, 2048, 2048, 128), (24, 16, 729, 729, 72), (65, 16, 257, 257, 80)]
for shape in shapes:
    row = "%-28s" % str(shape)
    for mode in os.environ.get("MODES", "auto,a32,narrow,wide").split(","):
        for k in ("OVO_ATTN_NARROW", "OVO_ATTN_WIDE", "OVO_ATTN_NO_TINY", "OVO_ATTN32"): os.environ.pop(k, None)
        if mode == "a32": os.environ["OVO_ATTN32"] = "1"
        if mode in ("narrow", "wide", "notiny"): os.environ["OVO_ATTN32"] = "0"
        if mode == "narrow": os.environ["OVO_ATTN_NARROW"] = "1"
        if mode == "wide": os.environ["OVO_ATTN_WIDE"] = "1"
        if mode == "notiny": os.environ["OVO_ATTN_NO_TINY"] = "1"
        us, tf, gbs = run(*shape)
        row += "  %s %8.1fus %5.0fTF %5.0fGB/s" % (mode, us, tf, gbs)
    print(row)
