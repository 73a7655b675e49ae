"""How much do the two encoder chains overlap when they share the chip?  ViT alone, SAM2 encoder alone, both on two streams.
Diagnosis tool (the frame pipeline runs them concurrently)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ovo_amd.pipeline import FramePipeline, synthetic_frames
dev = torch.device("cuda", 0)
pipe = FramePipeline(dev, n_map=100_000, extra_capacity=500_000)
f = synthetic_frames(1, dev)[0]
img = f.rgb.permute(2, 0, 1).contiguous()
tr = pipe.clip.textregion
x_sam = pipe.sam.preprocess(img)
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
def vit(): tr.get_img_features(img, scale=1 / 255.0)
def sam(): pipe.sam.forward(x_sam)
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn_n(fn, n); torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n
def fn_n(fn, n):
    for _ in range(n): fn()
def both():
    with torch.cuda.stream(s1): vit()
    with torch.cuda.stream(s2): sam()
tv, ts, tb = timed(vit), timed(sam), timed(both)
print(f"ViT alone {tv:.3f} ms   SAM2 encoder alone {ts:.3f} ms   sum {tv + ts:.3f}   both on two streams {tb:.3f} ms per pair   overlap factor {(tv + ts) / tb:.2f}")
