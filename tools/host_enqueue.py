"""Host enqueue time vs device time of the two encoder forwards (is the frame launch-bound?).  Diagnosis tool."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from ovo_amd.pipeline import FramePipeline, synthetic_frames
dev = torch.device("cuda", 0)
pipe = FramePipeline(dev, extra_capacity=2_000_000)
frames = synthetic_frames(6, dev)
for f in frames[:3]: pipe.step(f)
torch.cuda.synchronize()
f = frames[3]
x = pipe.sam.preprocess(f.rgb.permute(2, 0, 1).contiguous())
def measure(name, fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n): fn()
    t_host = time.perf_counter() - t0
    e1.record(); torch.cuda.synchronize()
    print(f"{name:28s} host enqueue {1e3 * t_host / n:7.3f} ms   device {e0.elapsed_time(e1) / n:7.3f} ms")
measure("sam.forward", lambda: pipe.sam.forward(x))
img = f.rgb.permute(2, 0, 1).contiguous()
tr = pipe.clip.textregion if hasattr(pipe.clip, "textregion") else None
measure("clip.extract (ViT+pool)", lambda: pipe.clip.extract_clip(f.rgb, f.masks, return_all=False) if tr is None else tr.predict(img, f.masks, scale=1 / 255.0))
t0 = time.perf_counter()
for g in frames[3:6]: pipe.step(g)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
print(f"step: host returns after {1e3 * t_host / 3:.3f} ms/frame; with final sync {1e3 * (time.perf_counter() - t0) / 3:.3f} ms/frame")
