"""Host micro-timings of the per-keyframe camera set-up (geometry_utils) on this machine's CPU."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from ovo_amd.utils import geometry_utils as G
torch.set_num_threads(1)
K = torch.tensor([[577.8, 0, 318.9], [0, 578.7, 242.7], [0, 0, 1]], dtype=torch.float32)
P = torch.eye(4); P[:3, 3] = torch.tensor([1.0, 2.0, 0.5])
def t(name, fn, n=2000):
    fn(); t0 = time.perf_counter()
    for _ in range(n): fn()
    print(f"{name:44s} {(time.perf_counter() - t0) / n * 1e6:8.1f} us")
c = G.frustum_corners_from_range(0.5, 3.0, 456, 616, P, K)
t("frustum_corners_from_range", lambda: G.frustum_corners_from_range(0.5, 3.0, 456, 616, P, K))
t("compute_camera_frustum_planes", lambda: G.compute_camera_frustum_planes(c))
t("compute_frustum_aabb", lambda: G.compute_frustum_aabb(c))
t("torch.linalg.inv(4x4)", lambda: torch.linalg.inv(P))
w2c = torch.linalg.inv(P)
t("make_camera (corners given)", lambda: G.make_camera(c, w2c, K, 0.05, 456, 616))
i = [0]
def miss():
    i[0] += 1
    return G.frame_camera(0.5 + 1e-6 * i[0], 3.0, 456, 616, P, K, 0.05)
t("frame_camera (miss)", miss)
t("frame_camera (hit)", lambda: G.frame_camera(0.5, 3.0, 456, 616, P, K, 0.03))
t("_cpu32(P)", lambda: G._cpu32(P))
t("P.numpy().tobytes()", lambda: P.numpy().tobytes())
