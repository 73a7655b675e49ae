"""The two batched resize launches of an encoder look-ahead group (14 frames): 28 TextRegion crops 640x480 -> 336^2 and 14 SAM2 inputs -> 1024^2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ovo_amd.encoders.hiera import SPECS as HS, HipHiera
from ovo_amd.encoders.vit import SPECS as VS, HipViT
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 14
frames = [(torch.rand(480, 640, 3, device=dev) * 255).to(torch.uint8) for _ in range(B)]
sam = object.__new__(HipHiera); sam.spec, sam.device = HS["hiera_b+"], dev
vit = object.__new__(HipViT); vit.spec, vit.device = VS["PE-Core-L14-336"], dev
o1 = torch.empty(B, 3, 1024, 1024, device=dev); o2 = torch.empty(2 * B, 3, 336, 336, device=dev)
crops = [(0, 0, 480, 640), (72, 152, 336, 336)]
def t(fn, n=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n
print("SAM2 %d x 1024^2 from HWC u8, one launch: %.1f us" % (B, t(lambda: HipHiera.preprocess_batch(sam, frames, out=o1))))
print("ViT %d x 2 crops 336^2 from HWC u8, one launch: %.1f us" % (B, t(lambda: HipViT.preprocess_batch(vit, frames, crops, scale=1 / 255.0, out=o2))))
