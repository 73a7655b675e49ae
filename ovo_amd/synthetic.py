"""Deterministic synthetic RGB-D stream, masks and text vectors for tests and bench.py.

The reference ships no frames: every workload here is generated from a seed.
Geometry is *consistent across frames*: depth is ray-cast from an analytic box room, so points
back-projected at frame t re-project onto frame t+1's depth within the match threshold, which is what
makes the match / vote / fuse stages do real work.

Layout follows the reference's ScanNet loader (ovo/entities/datasets.py:108-126): colour u8[H,W,3],
depth f32[h,w] in metres with 0 = invalid, c2w f32[4,4].
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np

# ScanNet camera (data/working/configs/ScanNet/scannet.yaml:3-11), crop_edge already applied.
SCANNET = dict(H=480, W=640, fx=577.590698, fy=578.729797, cx=318.905426, cy=242.683609, crop_edge=12)


def scannet_intrinsics(scale: float = 1.0) -> np.ndarray:
    """3x3 f32 intrinsics after crop_edge (datasets.py:32-41), optionally scaled for small tests."""
    k = np.eye(3, dtype=np.float32)
    k[0, 0] = SCANNET["fx"] * scale
    k[1, 1] = SCANNET["fy"] * scale
    k[0, 2] = (SCANNET["cx"] - SCANNET["crop_edge"]) * scale
    k[1, 2] = (SCANNET["cy"] - SCANNET["crop_edge"]) * scale
    return k


def scannet_depth_hw(scale: float = 1.0) -> Tuple[int, int]:
    e = SCANNET["crop_edge"]
    return int(round((SCANNET["H"] - 2 * e) * scale)), int(round((SCANNET["W"] - 2 * e) * scale))


def pose(frame: int, dx: float = 0.1, yaw: float = 0.3) -> np.ndarray:
    """c2w for frame t: +dx m along world x and +yaw rad about the camera's y axis per frame."""
    a = yaw * frame
    c, s = math.cos(a), math.sin(a)
    m = np.eye(4, dtype=np.float32)
    m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, s, -s, c
    m[0, 3] = dx * frame
    return m


ROOM_MIN = np.array([-2.0, -1.5, -2.2], dtype=np.float64)
ROOM_MAX = np.array([2.4, 1.5, 2.6], dtype=np.float64)


def render_depth(c2w: np.ndarray, k: np.ndarray, h: int, w: int, seed: int = 0,
                 invalid_frac: float = 0.05, noise: float = 0.004) -> np.ndarray:
    """z-depth of the box room seen from c2w; `invalid_frac` of the pixels are zeroed."""
    v, u = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    d_cam = np.stack([(u - k[0, 2]) / k[0, 0], (v - k[1, 2]) / k[1, 1], np.ones_like(u)], -1)
    r = c2w[:3, :3].astype(np.float64)
    o = c2w[:3, 3].astype(np.float64)
    d_w = d_cam @ r.T
    with np.errstate(divide="ignore", invalid="ignore"):
        t_hi = (ROOM_MAX - o) / d_w
        t_lo = (ROOM_MIN - o) / d_w
    t = np.where(d_w > 0, t_hi, t_lo)
    t = np.where(d_w == 0, np.inf, t).min(-1)
    rng = np.random.default_rng(1000 + seed)
    z = t + rng.normal(0.0, noise, size=t.shape)
    z[rng.random(t.shape) < invalid_frac] = 0.0
    return z.astype(np.float32)


def render_rgb(h: int, w: int, seed: int = 0) -> np.ndarray:
    """u8[h,w,3]: smooth gradient + noise, so resize paths see structure."""
    rng = np.random.default_rng(2000 + seed)
    v, u = np.meshgrid(np.linspace(0, 1, h), np.linspace(0, 1, w), indexing="ij")
    base = np.stack([u, v, 0.5 + 0.5 * np.sin(6.0 * (u + v) + seed)], -1) * 200.0
    img = base + rng.uniform(0, 55, size=(h, w, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


def frame(t: int, scale: float = 1.0, seed: int = 0):
    """One reference-style frame tuple (index, rgb, depth, c2w) (datasets.py:79)."""
    h, w = scannet_depth_hw(scale)
    k = scannet_intrinsics(scale)
    c2w = pose(t)
    return t, render_rgb(h, w, seed + t), render_depth(c2w, k, h, w, seed + t), c2w


def make_masks(h: int, w: int, grid=(4, 6), n_blobs: int = 8, seed: int = 0) -> np.ndarray:
    """bool[N,h,w]: grid cells plus overlapping axis-aligned blobs (SURVEY.md §8d 'Masks')."""
    rng = np.random.default_rng(3000 + seed)
    gh, gw = grid
    out = []
    ys = np.linspace(0, h, gh + 1).astype(int)
    xs = np.linspace(0, w, gw + 1).astype(int)
    for i in range(gh):
        for j in range(gw):
            m = np.zeros((h, w), dtype=bool)
            m[ys[i]:ys[i + 1], xs[j]:xs[j + 1]] = True
            out.append(m)
    for _ in range(n_blobs):
        bh, bw = rng.integers(h // 8, h // 3), rng.integers(w // 8, w // 3)
        y0, x0 = rng.integers(0, h - bh), rng.integers(0, w - bw)
        m = np.zeros((h, w), dtype=bool)
        m[y0:y0 + bh, x0:x0 + bw] = True
        out.append(m)
    return np.stack(out)


def masks_to_segmap(masks: np.ndarray) -> np.ndarray:
    """i32[h,w]: earlier mask wins, -1 = unassigned (the painting rule of segment_utils.py:12-27)."""
    seg = np.full(masks.shape[1:], -1, dtype=np.int32)
    for i, m in enumerate(masks):
        seg[m & (seg < 0)] = i
    return seg


def unit_vectors(n: int, d: int, seed: int = 0) -> np.ndarray:
    rng = np.random.default_rng(4000 + seed)
    x = rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


def padded_map(n_points: int, frames: int = 4, scale: float = 1.0, seed: int = 0) -> np.ndarray:
    """f32[n,3] map: back-projected synthetic frames padded with uniform points in the room box."""
    h, w = scannet_depth_hw(scale)
    k = scannet_intrinsics(scale).astype(np.float64)
    pts = []
    for t in range(frames):
        c2w = pose(t)
        z = render_depth(c2w, k, h, w, seed + t).astype(np.float64)[::2, ::2]
        v, u = np.meshgrid(np.arange(0, h, 2), np.arange(0, w, 2), indexing="ij")
        ok = z > 0
        x = (u[ok] - k[0, 2]) * z[ok] / k[0, 0]
        y = (v[ok] - k[1, 2]) * z[ok] / k[1, 1]
        p = np.stack([x, y, z[ok], np.ones_like(x)], -1) @ c2w.astype(np.float64).T
        pts.append(p[:, :3])
    pts = np.concatenate(pts).astype(np.float32)
    if pts.shape[0] >= n_points:
        return np.ascontiguousarray(pts[:n_points])
    rng = np.random.default_rng(5000 + seed)
    fill = rng.uniform(ROOM_MIN, ROOM_MAX, size=(n_points - pts.shape[0], 3)).astype(np.float32)
    return np.ascontiguousarray(np.concatenate([pts, fill]))
