"""BASELINE.json config 5: a fused fp16 map x 1k text embeddings -> (scores +) argmax.  Default: the WHOLE 10 M-point map on one GPU (15 GB of
f16 features; `1250000` = one GPU's share of an 8-GPU job).  Prints time per query, the MFMA rate (2 N Q D flops) against the dense f16 peak and
the algorithmic HBM rate ((N D + 8 N) x 2 B for the fused-argmax form, which never writes the score matrix).  usage: query_bench.py [N] [Q] [D]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ovo_amd.utils import clip_utils

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
q = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
d = int(sys.argv[3]) if len(sys.argv) > 3 else 768
F = torch.empty(n, d, device="cuda", dtype=torch.float16)
for s0 in range(0, n, 1 << 20):                                  # in slices: no 30 GB f32 temporary at 10 M points
    F[s0:s0 + (1 << 20)] = torch.nn.functional.normalize(torch.randn(min(1 << 20, n - s0), d, device="cuda"), dim=1).half()
T = torch.nn.functional.normalize(torch.randn(q, d, device="cuda"), dim=1)
flops = 2.0 * n * q * d
cases = ((True, "f32 scores + classes", n * d * 2 + n * q * 4 + n * 20, None), (True, "f16 scores + classes", n * d * 2 + n * q * 2 + n * 20, torch.float16),
         (False, "classes only        ", (n * d + 8 * n) * 2, None))
if n * q * 4 > 20e9:
    cases = cases[2:]                                             # the 40 GB score matrix of the full map is never wanted: classes only
for want_sim, label, bytes_, sdt in cases:
    for _ in range(2):
        clip_utils.similarity(F, T, want_argmax=True, want_sim=want_sim, sim_dtype=sdt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        sim, cls, conf = clip_utils.similarity(F, T, want_argmax=True, want_sim=want_sim, sim_dtype=sdt)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"query {n} x {d} f16  x  {q} texts, {label}: {ms:.2f} ms   {flops / ms / 1e9:.0f} TFLOP/s   {bytes_ / ms / 1e6:.0f} GB/s algorithmic   "
          f"({n / ms / 1e3:.1f} Mpoints/s)   MFMA frac {flops / ms / 1e9 / 2500:.3f} of 2.5 PFLOP/s, HBM frac {bytes_ / ms / 1e6 / 8000:.3f} of 8 TB/s")
