"""Oracle: frustum culling, projection, depth matching, back-projection (numpy in / numpy out).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Bulk per-point arithmetic lives in ovo_oracle.c with a
defined fused-multiply-add order; the tiny 8-corner / 6-plane host math uses torch-CPU ops, the same
library the reference runs, because ulp-level agreement of those 8x4 products cannot be restated
portably.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def frustum_corners(depth: np.ndarray, pose: np.ndarray, K: np.ndarray) -> np.ndarray:
    """geometry_utils.py:99-129 compute_camera_frustum_corners -> f32[8,3] world-frame corners."""
    h, w = depth.shape
    valid = depth[depth > 0]
    near, far = np.float32(valid.min()), np.float32(valid.max())
    px = torch.tensor([0, w, 0, w] * 2, dtype=torch.float32)
    py = torch.tensor([0, 0, h, h] * 2, dtype=torch.float32)
    z = torch.tensor([near] * 4 + [far] * 4, dtype=torch.float32)
    Kt = torch.from_numpy(_f32(K))
    x = (px - Kt[0, 2]) * z / Kt[0, 0]
    y = (py - Kt[1, 2]) * z / Kt[1, 1]
    cam = torch.stack([x, y, z, torch.ones(8)], dim=1)
    world = torch.einsum("ij,mj->mi", torch.from_numpy(_f32(pose)), cam)
    return world[:, :3].numpy().copy()


_PLANE_EDGES = ((2, 0, 1, 0), (6, 4, 5, 4), (4, 0, 2, 0), (7, 3, 1, 3), (5, 1, 3, 1), (6, 2, 0, 2))


def frustum_planes(corners: np.ndarray) -> np.ndarray:
    """geometry_utils.py:163-202 compute_camera_frustum_planes -> f32[6,4] (a,b,c,d).

    Plane i = cross(c[a]-c[b], c[c]-c[d]); its offset uses corner *i* (the reference enumerates the
    planes and indexes the corner list with the same counter, :201)."""
    c = torch.from_numpy(_f32(corners))
    normals = torch.stack([torch.linalg.cross(c[a] - c[b], c[e] - c[f]) for a, b, e, f in _PLANE_EDGES])
    d = torch.stack([-torch.dot(n, c[i]) for i, n in enumerate(normals)])
    return torch.cat([normals, d[:, None]], dim=1).float().numpy().copy()


def frustum_aabb(corners: np.ndarray) -> np.ndarray:
    """geometry_utils.py:205-215 -> f32[6] = (min xyz, max xyz)."""
    c = _f32(corners)
    return np.concatenate([c.min(0), c.max(0)]).astype(np.float32)


def frustum_point_ids(pts: np.ndarray, corners: np.ndarray) -> np.ndarray:
    """geometry_utils.py:252-276 compute_frustum_point_ids -> ascending i64 indices."""
    pts = _f32(pts)
    n = pts.shape[0]
    if n == 0:
        return np.zeros(0, np.int64)
    aabb, planes = frustum_aabb(corners), frustum_planes(corners)
    out = np.empty(n, np.int64)
    c = lib().orc_frustum_ids(_p(pts), n, _p(aabb), _p(planes), _p(out))
    return out[:c].copy()


def project(pts: np.ndarray, K: np.ndarray, w2c: np.ndarray) -> np.ndarray:
    """geometry_utils.py:26-43 project_3d_points (with w2c) -> i32[n,2] (u,v)."""
    pts = _f32(pts)
    out = np.empty((pts.shape[0], 2), np.int32)
    lib().orc_project(_p(pts), pts.shape[0], pts.shape[1], _p(_f32(w2c)), _p(_f32(K)), _p(out))
    return out


def match(depth: np.ndarray, w2c: np.ndarray, pts: np.ndarray, K: np.ndarray, th: float):
    """geometry_utils.py:46-89 match_3d_points_to_2d_pixels -> (i64[M] point index, i32[M,2] (u,v))."""
    pts, depth = _f32(pts), _f32(depth)
    n = pts.shape[0]
    idx = np.empty(n, np.int64)
    uv = np.empty((n, 2), np.int32)
    c = lib().orc_match(_p(pts), n, pts.shape[1] if n else 3, _p(_f32(w2c)), _p(_f32(K)), _p(depth),
                        depth.shape[0], depth.shape[1], float(th), _p(idx), _p(uv))
    return idx[:c].copy(), uv[:c].copy()


def gaussian_kernel1d(k: int = 7, sigma: float = 2.5) -> np.ndarray:
    """torchvision _get_gaussian_kernel1d: pdf at linspace(-(k-1)/2, (k-1)/2, k), normalised (fp32)."""
    half = (k - 1) * 0.5
    x = torch.linspace(-half, half, k)
    pdf = torch.exp(-0.5 * (x / sigma) ** 2)
    return (pdf / pdf.sum()).numpy()


def depth_filter(depth: np.ndarray, k: int = 7, sigma: float = 2.5, th: float = 0.05) -> np.ndarray:
    """geometry_utils.py:92-96: |d - blur(d)| > th -> -1.  blur = torchvision gaussian_blur (7x7,
    sigma 2.5, reflect padding).  torchvision is absent here: this restates its documented algorithm
    ("parity unpinned" for the blur itself); accumulation in float64, rounded once to fp32."""
    d = depth.astype(np.float64)
    p = k // 2
    k1 = gaussian_kernel1d(k, sigma).astype(np.float64)
    pad = np.pad(d, p, mode="reflect")
    h, w = d.shape
    acc = np.zeros_like(d)
    for dy in range(k):
        for dx in range(k):
            acc += (k1[dy] * k1[dx]) * pad[dy:dy + h, dx:dx + w]
    low = acc.astype(np.float32)
    hi = np.abs(depth - low)
    return np.where(hi > np.float32(th), np.float32(-1), depth).astype(np.float32)


def backproject(depth, rgb, explained, K, c2w, erode: bool, ds: int = 2):
    """vanilla_mapper.py:46-85 unproject part -> (f32[m,3] world points, u8[m,3] colours)."""
    depth = _f32(depth)
    h, w = depth.shape
    cap = ((h + ds - 1) // ds) * ((w + ds - 1) // ds)
    xyz = np.empty((cap, 3), np.float32)
    col = np.empty((cap, 3), np.uint8)
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    ex = None if explained is None else np.ascontiguousarray(explained, dtype=np.uint8)
    c = lib().orc_backproject(_p(depth), _p(rgb), None if ex is None else _p(ex), h, w, int(erode), ds,
                              _p(_f32(K)), _p(_f32(c2w)), _p(xyz), _p(col))
    return xyz[:c].copy(), col[:c].copy()
