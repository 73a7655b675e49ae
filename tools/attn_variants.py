"""Resident attention kernel: workgroup size x query splits (OVO_ATTN_RES_THREADS / OVO_ATTN_RES_SPLITS), one process per variant."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = '''
import ctypes as C, os, sys
sys.path.insert(0, %r)
import torch
from ovo_amd import _lib as L
dev = torch.device("cuda", 0); lib = L.load()
def run(B, H, Tq, Tk, hd, iters=50):
    D = H * hd; T = max(Tq, Tk)
    qkv = torch.randn(B, T, 3, H, hd, device=dev).to(torch.bfloat16)
    out = torch.zeros(B, Tq, D, dtype=torch.bfloat16, device=dev)
    a = L.Attention(); base = qkv.data_ptr()
    a.q, a.k, a.v, a.o = base, base + D * 2, base + 4 * D, out.data_ptr()
    a.q_sb = a.k_sb = a.v_sb = T * 3 * D; a.q_sh = a.k_sh = a.v_sh = hd; a.q_st = a.k_st = a.v_st = 3 * D
    a.o_sb, a.o_sh, a.o_st = Tq * D, hd, D
    a.B, a.H, a.Tq, a.Tk, a.hd, a.scale = B, H, Tq, Tk, hd, hd ** -0.5
    for _ in range(5): L.check(lib.ovo_attention(C.byref(a), L.stream()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(iters): L.check(lib.ovo_attention(C.byref(a), L.stream()))
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters
print("  ".join("%%s %%.1f us" %% (s, run(*s)) for s in [(300, 8, 196, 196, 56), (300, 16, 49, 196, 56), (24, 16, 577, 577, 64)]))
''' % ROOT
for threads in (0, 256, 512):
    for splits in (0, 1, 2, 3, 4):
        env = dict(os.environ)
        if threads: env["OVO_ATTN_RES_THREADS"] = str(threads)
        if splits: env["OVO_ATTN_RES_SPLITS"] = str(splits)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print(f"threads {threads or 'auto'} splits {splits or 'auto'}: {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]}", flush=True)
