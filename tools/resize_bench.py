"""ovo_resize_normalize on the bench's three per-frame calls: 640x480 HWC u8 -> 1024^2 (SAM2) and two 480x320 crops -> 336^2 (ViT)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ovo_amd.encoders.hiera import SPECS as HS, HipHiera
from ovo_amd.encoders.vit import SPECS as VS, HipViT
dev = torch.device("cuda", 0)
img = (torch.rand(480, 640, 3, device=dev) * 255).to(torch.uint8)
sam = object.__new__(HipHiera); sam.spec, sam.device = HS["hiera_b+"], dev
vit = object.__new__(HipViT); vit.spec, vit.device = VS["PE-Core-L14-336"], dev
o1 = torch.empty(1, 3, 1024, 1024, device=dev); o2 = torch.empty(2, 3, 336, 336, device=dev)
crops = [(0, 0, 480, 320), (0, 320, 480, 320)]
def t(fn, n=200):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n
print("SAM2 1024^2 from HWC u8: %.1f us" % t(lambda: HipHiera.preprocess(sam, img, out=o1)))
print("ViT 2 crops 336^2 from HWC u8: %.1f us (both)" % t(lambda: HipViT.preprocess(vit, img, crops, scale=1 / 255.0, out=o2)))
