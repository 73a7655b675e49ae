"""Marginal cost of each part of the frame: the bench loop with parts switched off.  Diagnosis tool."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ovo_amd.pipeline import FramePipeline, synthetic_frames
dev = torch.device("cuda", 0)
frames = synthetic_frames(45, dev)
def run(**kw):
    pipe = FramePipeline(dev, n_map=1_000_000, extra_capacity=3_000_000, **kw)
    it = iter(frames)
    for _ in range(5): pipe.step(next(it))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40): pipe.step(next(it))
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 40
    del pipe
    torch.cuda.empty_cache()
    return ms
for name, kw in (("full", {}), ("no SAM2 encoder", {"sam_card": None}), ("no dense fusion/query", {"dense": False}),
                 ("no SAM2, no dense", {"sam_card": None, "dense": False})):
    print(f"{name:26s} {run(**kw):6.3f} ms/frame")
