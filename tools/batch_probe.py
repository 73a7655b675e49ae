"""Where a frame's time goes with encoder look-ahead batching: the two encoders alone / together at batch B, and the frame loop
with parts switched off.  Diagnosis tool."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ovo_amd.pipeline import FramePipeline, synthetic_frames
dev = torch.device("cuda", 0)
frames = synthetic_frames(56, dev)
def timed(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n
pipe = FramePipeline(dev, n_map=100_000, extra_capacity=500_000)
tr = pipe.clip.textregion
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
for B in (1, 2, 4, 8):
    imgs = [f.rgb.permute(2, 0, 1).contiguous() for f in frames[:B]]
    crops = tr._crops(*imgs[0].shape[1:])
    xb = torch.cat([tr.vlm.preprocess(i, crops, scale=1 / 255.0) for i in imgs])
    xs = torch.cat([pipe.sam.preprocess(i) for i in imgs])
    vit = lambda: tr.vlm.forward(xb, tokens=True)
    sam = lambda: pipe.sam.forward(xs)
    def both():
        with torch.cuda.stream(s1): vit()
        with torch.cuda.stream(s2): sam()
    tv, ts, tb = timed(vit), timed(sam), timed(both)
    print(f"B={B}: ViT {tv / B:.3f}  SAM2 {ts / B:.3f}  sum {(tv + ts) / B:.3f}  both on two streams {tb / B:.3f} ms per frame")
del pipe
torch.cuda.empty_cache()
def run(B, **kw):
    pipe = FramePipeline(dev, n_map=1_000_000, extra_capacity=3_000_000, encoder_batch=B, **kw)
    pos = 0
    def go(n):
        nonlocal pos
        end = pos + n
        for _ in range(n):
            pipe.step(frames[pos], frames[pos + 1:end]); pos += 1
    go(8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    go(40)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / 40
    del pipe
    torch.cuda.empty_cache()
    return ms
for B in (1, 4, 8):
    for name, kw in (("full", {}), ("no SAM2 encoder", {"sam_card": None}), ("no dense fusion/query", {"dense": False}),
                     ("no SAM2, no dense", {"sam_card": None, "dense": False})):
        print(f"B={B} {name:26s} {run(B, **kw):6.3f} ms/frame")
