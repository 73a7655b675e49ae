"""The LayerNorm fold's pieces alone, at the headline ViT's shapes (PE-L/14-336: 28 crops x 577 tokens = 16156 rows, width 1024, mlp 4096):
    plain   LayerNorm kernel (f32 x -> bf16 h) + product          |  out-projection / FC2 with f32 += residual
    folded  ovo_gemm_fold_in (A = bf16 x, statistics per row)     |  ovo_gemm_fold_out (the same + bf16 copy + partial statistics)
and one block's six launches in sequence (attention left out) in both forms.   python tools/fold_bench.py [rows]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ovo_amd import _lib as L

M = int(sys.argv[1]) if len(sys.argv) > 1 else 16156
D, MLP, T, HD = 1024, 4096, 577, 64
dev = torch.device("cuda", 0)
lib = L.load()
g0 = torch.Generator().manual_seed(0)


def rn(*shape, std=1.0, dtype=torch.float32):
    return (torch.randn(*shape, generator=g0) * std).to(dev, dtype)


x = rn(M, D)
h = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
xb = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
stats = torch.zeros(16, M, 2, device=dev)
att = rn(M, D, dtype=torch.bfloat16)
u = torch.empty(M, MLP, dtype=torch.bfloat16, device=dev)
qkv = torch.empty(M, 3 * D, dtype=torch.bfloat16, device=dev)
gamma, beta = 1 + rn(D, std=0.1), rn(D, std=0.1)
Wq, bq = rn(3 * D, D, std=D ** -0.5), rn(3 * D, std=0.02)
W1, b1 = rn(MLP, D, std=D ** -0.5), rn(MLP, std=0.02)
Wo, bo = rn(D, D, std=D ** -0.5, dtype=torch.bfloat16), rn(D, std=0.02)
W2, b2 = rn(D, MLP, std=MLP ** -0.5, dtype=torch.bfloat16), rn(D, std=0.02)
Wqf, bqf, csq = [t.to(dev) for t in L.fold_layernorm(Wq.cpu(), bq.cpu(), gamma.cpu(), beta.cpu())]
W1f, b1f, cs1 = [t.to(dev) for t in L.fold_layernorm(W1.cpu(), b1.cpu(), gamma.cpu(), beta.cpu())]
Wq_b, W1_b, Wqf_b, W1f_b = Wq.bfloat16(), W1.bfloat16(), Wqf.bfloat16(), W1f.bfloat16()
cos, sin = rn(T, HD), rn(T, HD)
rope = L.Rope(cos.data_ptr(), sin.data_ptr(), T, HD, 2 * D, 1)


def desc(a, w, bias, out, add=None, act=0):
    g = L.Gemm()
    g.A, g.lda, g.W, g.ldw, g.bias = a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), bias.data_ptr()
    g.C, g.ldc = out.data_ptr(), out.stride(0)
    g.add, g.ld_add = (add.data_ptr(), add.stride(0)) if add is not None else (None, 0)
    g.M, g.N, g.K = a.shape[0], w.shape[0], a.shape[1]
    g.in_dtype, g.out_dtype, g.act, g.alpha = 2, L.DTYPE_CODE[out.dtype], act, 1.0
    return g


def ln():
    L.check(lib.ovo_layernorm(L.ptr(x), D, M, D, L.ptr(gamma), L.ptr(beta), 1e-5, L.ptr(h), D, 2, L.stream()))


d_qkv, d_fc1 = desc(h, Wq_b, bq, qkv), desc(h, W1_b, b1, u, act=1)
d_qkvf, d_fc1f = desc(xb, Wqf_b, bqf, qkv), desc(xb, W1f_b, b1f, u, act=1)
d_out, d_fc2 = desc(att, Wo, bo, x, add=x), desc(u, W2, b2, x, add=x)
ops = {
    "LayerNorm kernel": ln,
    "QKV + rope (plain)": lambda: L.check(lib.ovo_gemm_rope(C.byref(d_qkv), C.byref(rope), L.stream())),
    "QKV + rope (folded)": lambda: L.check(lib.ovo_gemm_fold_in(C.byref(d_qkvf), C.byref(rope), L.ptr(stats), M, 16, D, L.ptr(csq), 1e-5, L.stream())),
    "FC1 + GELU (plain)": lambda: L.check(lib.ovo_gemm(C.byref(d_fc1), L.stream())),
    "FC1 + GELU (folded)": lambda: L.check(lib.ovo_gemm_fold_in(C.byref(d_fc1f), None, L.ptr(stats), M, 16, D, L.ptr(cs1), 1e-5, L.stream())),
    "out projection (plain)": lambda: L.check(lib.ovo_gemm(C.byref(d_out), L.stream())),
    "out projection (+ bf16 copy + statistics)": lambda: L.check(lib.ovo_gemm_fold_out(C.byref(d_out), L.ptr(xb), D, L.ptr(stats), M, L.stream())),
    "FC2 (plain)": lambda: L.check(lib.ovo_gemm(C.byref(d_fc2), L.stream())),
    "FC2 (+ bf16 copy + statistics)": lambda: L.check(lib.ovo_gemm_fold_out(C.byref(d_fc2), L.ptr(xb), D, L.ptr(stats), M, L.stream())),
}


def block_plain():
    ln(); ops["QKV + rope (plain)"](); ops["out projection (plain)"](); ln(); ops["FC1 + GELU (plain)"](); ops["FC2 (plain)"]()


def block_fold():
    ops["QKV + rope (folded)"](); ops["out projection (+ bf16 copy + statistics)"](); ops["FC1 + GELU (folded)"](); ops["FC2 (+ bf16 copy + statistics)"]()


ops["block without attention, plain: 6 launches"] = block_plain
ops["block without attention, folded: 4 launches"] = block_fold


def time_it(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


L.check(lib.ovo_gemm_fold_stats(L.ptr(x), D, M, D, L.ptr(xb), D, L.ptr(stats), L.stream()))
ops["out projection (+ bf16 copy + statistics)"]()                 # statistics in the 16-partial layout the folded products read below
print(f"rows {M}, width {D}, mlp {MLP}; median (min) of 7 alternating rounds of 20 launches")
names = list(ops)
res = {n: [] for n in names}
for rnd in range(7):                                               # every form in every round: clock drift and neighbours hit all of them alike
    for n in names:
        x.normal_()
        res[n].append(time_it(ops[n], reps=20))
for n in names:
    v = sorted(res[n])
    print(f"{n:48s} {v[len(v) // 2]:8.2f} us  ({v[0]:.2f})")
