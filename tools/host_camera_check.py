"""Bit-equality of the batched frustum set-up (geometry_utils.frustum_corners_from_range / compute_camera_frustum_planes) with the
op-by-op form that mirrors the reference line by line, on random cameras, on THIS host's CPU (ATen picks its vector kernels per CPU)."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from ovo_amd.utils import geometry_utils as G
torch.set_num_threads(1)
PD = G._PLANE_DEF
def corners_ref(near, far, h, w, T, K):
    px = torch.tensor(G._CORNER_X * 2, dtype=torch.float32) * float(w)
    py = torch.tensor(G._CORNER_Y * 2, dtype=torch.float32) * float(h)
    z = torch.tensor([near] * 4 + [far] * 4, dtype=torch.float32)
    cam = torch.stack([(px - K[0, 2]) * z / K[0, 0], (py - K[1, 2]) * z / K[1, 1], z, torch.ones(8)], dim=1)
    return torch.einsum("ij,mj->mi", T, cam)[:, :3].contiguous()
def planes_ref(c):
    n = torch.stack([torch.linalg.cross(c[a] - c[b], c[e] - c[f]) for a, b, e, f in PD])
    d = torch.stack([-torch.dot(n[i], c[i]) for i in range(6)])
    return torch.cat([n, d[:, None]], dim=1).float()
rng = np.random.default_rng(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
okc = okp = 0
for _ in range(N):
    R = np.linalg.qr(rng.standard_normal((3, 3)))[0].astype(np.float32)
    T = np.eye(4, dtype=np.float32); T[:3, :3] = R; T[:3, 3] = (rng.standard_normal(3) * 3).astype(np.float32)
    K = torch.tensor([[500 + rng.random() * 100, 0, 320 + rng.random() * 5], [0, 500 + rng.random() * 100, 240 + rng.random() * 5], [0, 0, 1]], dtype=torch.float32)
    near, far = float(np.float32(0.3 + rng.random())), float(np.float32(2 + rng.random() * 5))
    Tt = torch.from_numpy(T)
    cr = corners_ref(near, far, 480, 640, Tt, K)
    cn = G.frustum_corners_from_range(near, far, 480, 640, Tt, K)
    okc += int(torch.equal(cr, cn))
    okp += int(torch.equal(planes_ref(cr), G.compute_camera_frustum_planes(cr)))
print(f"corners equal {okc}/{N}, planes equal {okp}/{N}")
t = time.perf_counter()
for _ in range(500): G.compute_camera_frustum_planes(G.frustum_corners_from_range(near, far, 480, 640, Tt, K))
print("new: %.1f us per camera" % ((time.perf_counter() - t) / 500 * 1e6))
t = time.perf_counter()
for _ in range(500): planes_ref(corners_ref(near, far, 480, 640, Tt, K))
print("ref: %.1f us per camera" % ((time.perf_counter() - t) / 500 * 1e6))
