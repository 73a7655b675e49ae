"""What does the vendor library (hipBLASLt through torch) reach on our GEMM shapes?  Headroom estimate only -- not a product path."""
import os; os.environ.setdefault("OVO_KNOBS_DYNAMIC", "1")    # this tool flips OVO_* knobs between launches
import torch
dev = "cuda"
shapes = [(1154, 3072, 1024), (1154, 1024, 1024), (1154, 4096, 1024), (1154, 1024, 4096), (4096, 1792, 448), (4096, 448, 1792), (4096, 1344, 448),
          (65536, 448, 128), (65536, 112, 448), (16384, 896, 224), (1250000, 1000, 768),
          (13848, 3072, 1024), (13848, 1024, 1024), (13848, 4096, 1024), (13848, 1024, 4096), (49152, 1792, 448), (49152, 448, 1792), (58800, 1344, 448),   # 12 frames per launch
          (4096, 4096, 4096), (8192, 8192, 8192)]
for m, n, k in shapes:
    a = torch.randn(m, k, device=dev, dtype=torch.bfloat16); w = torch.randn(n, k, device=dev, dtype=torch.bfloat16)
    for _ in range(5): torch.nn.functional.linear(a, w)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    it = 30 if m < 100000 else 5
    for _ in range(it): torch.nn.functional.linear(a, w)
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / it
    print(f"{(m, n, k)!s:24s} {us:9.1f} us {2.0 * m * n * k / us / 1e6:7.0f} TF")
