"""Known-byte-count workload for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on this box (MI355X_MICROARCH.md, HBM
section: "calibrate on a known byte count in your own access pattern before trusting an absolute").

    fill : writes exactly 1 GiB                (WRITE_SIZE calibration)
    copy : reads 1 GiB and writes 1 GiB        (FETCH_SIZE calibration, 16 B / lane streaming)
    ovo_similarity over 1M x 1024 f32 rows: reads 4 GiB  (our own MFMA streaming pattern)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ovo_amd.utils import clip_utils

n = 1 << 28
a = torch.empty(n, dtype=torch.float32, device="cuda")
b = torch.empty(n, dtype=torch.float32, device="cuda")
for _ in range(3):
    a.fill_(1.0)
    b.copy_(a)
F = a.view(1 << 18, 1024)                 # the same 1 GiB, read once by our MFMA streaming pattern
T = torch.randn(10, 1024, device="cuda")
for _ in range(3):
    clip_utils.similarity(F, T, want_sim=False, want_argmax=True)
torch.cuda.synchronize()
print("calib ok")
