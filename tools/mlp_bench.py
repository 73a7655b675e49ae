"""The fused MLP (mlp_stream.hip: x += fc2(GELU(fc1(LN(x))))) against the two launches it replaces, at Hiera stage-1 / stage-2 shapes of a 12-frame
group.  python tools/mlp_bench.py   (OVO_MLP_RB = row blocks per wave of the fused kernel)"""
import ctypes as C
import os
import sys

os.environ.setdefault("OVO_KNOBS_DYNAMIC", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ovo_amd import _lib as L

dev = torch.device("cuda", 0)
lib = L.load()


def run(rows, d, k1, iters=20):
    hid = 4 * d
    x = torch.randn(rows, d, device=dev)
    gamma, beta = torch.ones(d, device=dev), torch.zeros(d, device=dev)
    w1 = torch.zeros(hid, k1, dtype=torch.bfloat16, device=dev)
    w1[:, :d] = (torch.randn(hid, d, device=dev) * d ** -0.5).to(torch.bfloat16)
    w2 = (torch.randn(d, hid, device=dev) * hid ** -0.5 * 0.1).to(torch.bfloat16)
    b1, b2 = torch.zeros(hid, device=dev), torch.zeros(d, device=dev)
    h = torch.empty(rows, hid, dtype=torch.bfloat16, device=dev)

    def fused():
        L.check(lib.ovo_mlp_f32(x.data_ptr(), rows, d, gamma.data_ptr(), beta.data_ptr(), 1e-6, w1.data_ptr(), k1, b1.data_ptr(), hid, w2.data_ptr(), hid,
                                b2.data_ptr(), L.stream()))

    q = L.Gemm()
    q.A, q.lda, q.W, q.ldw, q.bias, q.C, q.ldc, q.add, q.ld_add = None, k1, w1.data_ptr(), k1, b1.data_ptr(), h.data_ptr(), hid, None, 0
    q.M, q.N, q.K, q.in_dtype, q.out_dtype, q.act, q.alpha = rows, hid, k1, 2, 2, 1, 1.0
    q2 = L.Gemm()
    q2.A, q2.lda, q2.W, q2.ldw, q2.bias, q2.C, q2.ldc, q2.add, q2.ld_add = h.data_ptr(), hid, w2.data_ptr(), hid, b2.data_ptr(), x.data_ptr(), d, x.data_ptr(), d
    q2.M, q2.N, q2.K, q2.in_dtype, q2.out_dtype, q2.act, q2.alpha = rows, d, hid, 2, 0, 0, 1.0

    def two():
        L.check(lib.ovo_gemm_f32a(C.byref(q), None, x.data_ptr(), d, gamma.data_ptr(), beta.data_ptr(), 1e-6, 1, 0, L.stream()))
        L.check(lib.ovo_gemm(C.byref(q2), L.stream()))
    out = {}
    legs = (("fused", fused),) if os.environ.get("ONLY_FUSED") else (("fused", fused), ("two launches", two))
    iters = int(os.environ.get("ITERS", iters))
    for name, fn in legs:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        out[name] = 1e3 * e0.elapsed_time(e1) / iters
    by = 8.0 * rows * d
    out.setdefault("two launches", float("nan"))
    print(f"({rows}, d {d}, hidden {hid}): fused {out['fused']:7.1f} us = {by / out['fused'] / 1e3:6.0f} GB/s of the stream in + out, "
          f"{4.0 * rows * hid * d / out['fused'] / 1e6:5.0f} TFLOP/s;  two launches {out['two launches']:7.1f} us")


for lut in ("1", ""):
    os.environ.pop("OVO_MLP_GELU_POLY", None)
    if not lut:
        os.environ["OVO_MLP_GELU_POLY"] = "1"
    for rb in os.environ.get("RBS", "0").split(","):          # variant number of mlp_stream_launch, 0 = the default instantiation
        os.environ["OVO_MLP_RB"] = rb
        print("GELU", "table" if lut else "polynomial", " OVO_MLP_RB =", rb)
        try:
            run(786432, 112, 128)
            run(196608, 224, 256)
        except L.OvoHipError as e:
            print("   (no such instantiation)")
