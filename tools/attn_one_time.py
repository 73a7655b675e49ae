"""One attention shape, timed: python tools/attn_one_time.py B H Tq Tk hd [iters]  (env knobs of ovo_attention apply)"""
import os; os.environ.setdefault("OVO_KNOBS_DYNAMIC", "1")    # this tool flips OVO_* knobs between launches
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ovo_amd import _lib as L
B, H, Tq, Tk, hd = (int(v) for v in sys.argv[1:6])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 20
dev = torch.device("cuda", 0); lib = L.load()
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
print("%s  %.1f us  %.0f TF" % ((B, H, Tq, Tk, hd), us, 4.0 * B * H * Tq * Tk * hd / us / 1e6))
