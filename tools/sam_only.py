"""SAM2 image encoder forward in a loop (for rocprofv3 --kernel-trace --stats).  Diagnosis tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ovo_amd.encoders.hiera import SPECS, HipHiera
dev = torch.device("cuda", 0)
sam = HipHiera(SPECS[os.environ.get("SAM", "hiera_b+")], None, dev, 0)
x = torch.randn(1, 3, 1024, 1024, device=dev)
for _ in range(3): sam.forward(x)
torch.cuda.synchronize()
n = int(os.environ.get("N", 20))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n): sam.forward(x)
e1.record(); torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1) / n:.3f} ms per forward")
