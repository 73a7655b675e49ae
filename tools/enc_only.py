"""One encoder alone, batched: python tools/enc_only.py {vit|sam} B [iters] -- for rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ovo_amd.encoders.hiera import SPECS as HS, HipHiera
from ovo_amd.encoders.vit import SPECS as VS, HipViT
which, B = sys.argv[1], int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda", 0)
if which == "vit":
    enc = HipViT(VS["PE-Core-L14-336"], None, dev, 0)
    x = torch.randn(2 * B, 3, 336, 336, device=dev)
    fn = lambda: enc.forward(x, tokens=True)
else:
    enc = HipHiera(HS[os.environ.get("SAM", "hiera_b+")], None, dev, 0)
    x = torch.randn(B, 3, 1024, 1024, device=dev)
    fn = lambda: enc.forward(x)
for _ in range(3): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters): fn()
e1.record(); torch.cuda.synchronize()
print(f"{which} B={B}: {e0.elapsed_time(e1) / iters / B:.3f} ms per frame")
