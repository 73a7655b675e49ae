import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from ovo_amd import _lib as L
dev = torch.device("cuda", 0)
lib = L.load()
m = torch.zeros(32, 480, 640, dtype=torch.uint8, device=dev)
d = torch.rand(456, 616, device=dev); o = torch.empty_like(d)
L.check(lib.ovo_depth_filter(L.ptr(d), 456, 616, 7, 2.5, 0.05, L.ptr(o), L.stream())); torch.cuda.synchronize()
pairs = torch.tensor([0, 1], dtype=torch.int32, device=dev); rows = torch.tensor([0], dtype=torch.int32, device=dev); area = torch.empty(1, dtype=torch.int32, device=dev)
for name, fn in (("mask_or", lambda: lib.ovo_mask_or(L.ptr(m), 480 * 640, L.ptr(pairs), 1, L.stream())),
                 ("mask_area", lambda: lib.ovo_mask_area(L.ptr(m), 480 * 640, L.ptr(rows), 1, L.ptr(area), L.stream())),
                 ("tolist", lambda: area.tolist()),
                 ("view_u8", lambda: torch.zeros(4, 4, dtype=torch.bool, device=dev).view(torch.uint8))):
    for i in range(3):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        print(name, i, round(1e3 * (time.perf_counter() - t0), 3), "ms")
