"""BASELINE.json config 5, one GPU's share: (10M / 8) x D fp16 fused map  x  1k text embeddings -> scores + argmax.
Prints time per query and the achieved MFMA / HBM rates.  Diagnosis tool, not the bench line."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ovo_amd.utils import clip_utils

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_250_000
q = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
d = int(sys.argv[3]) if len(sys.argv) > 3 else 768
F = torch.nn.functional.normalize(torch.randn(n, d, device="cuda"), dim=1).half()
T = torch.nn.functional.normalize(torch.randn(q, d, device="cuda"), dim=1)
flops = 2.0 * n * q * d
for want_sim, label, bytes_ in ((True, "scores + classes", n * d * 2 + n * q * 4 + n * 20), (False, "classes only    ", n * d * 2 + n * 20)):
    for _ in range(2):
        clip_utils.similarity(F, T, want_argmax=True, want_sim=want_sim)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        sim, cls, conf = clip_utils.similarity(F, T, want_argmax=True, want_sim=want_sim)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"query {n} x {d} f16  x  {q} texts, {label}: {ms:.2f} ms   {flops / ms / 1e9:.0f} TFLOP/s   {bytes_ / ms / 1e6:.0f} GB/s algorithmic   "
          f"({n / ms / 1e3:.1f} Mpoints/s)")
