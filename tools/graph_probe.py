"""Does a captured HIP graph of one encoder forward (fixed shapes, one keyframe: the online mode's ~380 dependent launches) replay faster than the
direct launches?  ViT PE-L/14-336 at 2 crops and hiera_b+ at 1 frame, each timed as direct calls and as torch.cuda.CUDAGraph replays.
python tools/graph_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ovo_amd.encoders import hiera as EH, vit as EV

dev = torch.device("cuda", 0)


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def probe(name, model, x, **kw):
    out = model.forward(x, **kw)                        # warm-up: workspaces, function attributes, first launches
    torch.cuda.synchronize()
    direct = timed(lambda: model.forward(x, **kw))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    try:
        with torch.cuda.stream(s):
            model.forward(x, **kw)
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g, stream=s):
            gout = model.forward(x, **kw)
        torch.cuda.synchronize()
        replay = timed(g.replay)
        a = out if torch.is_tensor(out) else out[-1]
        b = gout if torch.is_tensor(gout) else gout[-1]
        same = bool(torch.equal(a, b))
        print(f"{name}: direct {direct:.3f} ms, graph replay {replay:.3f} ms ({direct / replay:.2f}x), outputs equal: {same}")
    except Exception as e:
        print(f"{name}: direct {direct:.3f} ms, capture failed: {e!r}")


vit = EV.HipViT(EV.SPECS["PE-Core-L14-336"], None, dev, 0)
probe("ViT PE-L/14-336, 2 crops", vit, torch.randn(2, 3, 336, 336, device=dev), tokens=True)
vb = EV.HipViT(EV.SPECS["ViT-B-16-qg"], None, dev, 0) if "ViT-B-16-qg" in EV.SPECS else None
if vb is not None:
    probe("ViT-B/16, 1 image", vb, torch.randn(1, 3, 224, 224, device=dev))
sam = EH.HipHiera(EH.SPECS["hiera_b+"], None, dev, 0)
probe("hiera_b+, 1 frame", sam, torch.randn(1, 3, 1024, 1024, device=dev))
