"""The per-frame dense query of the bench workload: N x 1024 f32 accumulators (+ counts) x 10 texts, classes + confidences.  Diagnosis tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ovo_amd.utils import clip_utils
dev = torch.device("cuda", 0)
n, d, q = int(os.environ.get("N", 1_300_000)), 1024, 10
acc = torch.randn(n, d, device=dev)
cnt = torch.randint(0, 5, (n,), device=dev, dtype=torch.int32)
txt = torch.nn.functional.normalize(torch.randn(q, d, device=dev), dim=-1)
for _ in range(3): clip_utils.similarity(acc, txt, cnt=cnt, want_sim=False, want_argmax=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(20): out = clip_utils.similarity(acc, txt, cnt=cnt, want_sim=False, want_argmax=True)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"dense query {n} x {d} f32 x {q} texts: {ms:.3f} ms   {n * d * 4 / ms / 1e9:.2f} TB/s")
