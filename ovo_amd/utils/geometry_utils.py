"""Frustum culling, projection, depth matching and the depth high-pass filter on MI355X.

Host-side mirror of the reference's `ovo/utils/geometry_utils.py` (same function names, argument meaning
and return layout) over libovo_hip.so.  The per-point arithmetic runs in HIP (csrc/geometry.hip); only the
8-corner / 6-plane set-up, a few dozen flops per frame, stays on the host in torch-CPU so that it is the
same arithmetic the reference's CPU path performs (DESIGN.md "bit-exactness").

Reference lines each function replaces are given in its docstring.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np
import torch

from .. import _lib as L

# corner order of the reference (near plane first): (0,0) (w,0) (0,h) (w,h), then the same at far depth
_CORNER_X = (0.0, 1.0, 0.0, 1.0)
_CORNER_Y = (0.0, 0.0, 1.0, 1.0)
# plane i = cross(c[a]-c[b], c[c]-c[d]); near, far, left, right, top, bottom
_PLANE_DEF = ((2, 0, 1, 0), (6, 4, 5, 4), (4, 0, 2, 0), (7, 3, 1, 3), (5, 1, 3, 1), (6, 2, 0, 2))


def _cpu32(t) -> torch.Tensor:
    if isinstance(t, np.ndarray):
        t = torch.from_numpy(t)
    return t.detach().to(device="cpu", dtype=torch.float32)


def to_device(arr, dtype: torch.dtype, device) -> torch.Tensor:
    """Frame arrays arrive as numpy (the reference's dataset tuples) or as tensors already resident on the GPU."""
    if isinstance(arr, torch.Tensor):
        t = arr if arr.dtype == dtype else arr.to(dtype)
        out = (t if t.is_cuda else t.to(device, non_blocking=True)).contiguous()
        if out is not arr and hasattr(arr, "_ovo_range") and getattr(arr, "_ovo_range_version", arr._version) == arr._version:
            out._ovo_range, out._ovo_range_version = arr._ovo_range, out._version
        return out
    np_dtype = {torch.float32: np.float32, torch.uint8: np.uint8, torch.int32: np.int32}[dtype]
    return torch.from_numpy(np.ascontiguousarray(arr, dtype=np_dtype)).to(device, non_blocking=True)


def depth_range(depth) -> Tuple[float, float]:
    """min / max of the valid (> 0) depths (inf, <= 0 when there is none).  numpy input costs no device sync; a device tensor is
    reduced once and the pair is remembered ON the tensor (`depth._ovo_range`): the mapper and the tracker both need it for the same
    frame (vanilla_mapper.py:60, ovo.py:209), and whoever uploads a frame from host memory can attach it for free (`tag_depth_range`)
    -- each device-side evaluation is a host sync in the middle of the keyframe."""
    if isinstance(depth, np.ndarray):
        v = depth[depth > 0]
        if v.size == 0:
            return float("inf"), 0.0
        return float(v.min()), float(v.max())
    cached = getattr(depth, "_ovo_range", None)
    if cached is not None and getattr(depth, "_ovo_range_version", depth._version) == depth._version:
        return cached                                  # (a buffer rewritten in place -- buf.copy_(next_depth) -- has a new version: recompute)
    big = torch.where(depth > 0, depth, torch.full_like(depth, float("inf"))).min()
    top = depth.max()
    lo, hi = torch.stack([big, top]).tolist()
    depth._ovo_range, depth._ovo_range_version = (lo, hi), depth._version
    return lo, hi


def tag_depth_range(depth_dev: torch.Tensor, depth_host: np.ndarray) -> torch.Tensor:
    """Attach the (min valid, max) range of a depth map, computed from its HOST copy, to the device tensor of the same frame."""
    depth_dev._ovo_range, depth_dev._ovo_range_version = depth_range(np.asarray(depth_host)), depth_dev._version
    return depth_dev


_corner_px = {}


def frustum_corners_from_range(near: float, far: float, h: int, w: int, pose, intrinsics) -> torch.Tensor:
    """8 world-frame frustum corners, f32[8,3] on the CPU (geometry_utils.py:99-129).  Same float32 arithmetic as the reference's
    op-by-op form, batched: the pose product is written as the left-to-right sum torch's einsum evaluates
    (tools/host_camera_check.py: bit-equal on random cameras on this host)."""
    K, T = _cpu32(intrinsics), _cpu32(pose)
    pxy = _corner_px.get((h, w))
    if pxy is None:
        pxy = _corner_px[(h, w)] = (torch.tensor(_CORNER_X * 2, dtype=torch.float32) * float(w), torch.tensor(_CORNER_Y * 2, dtype=torch.float32) * float(h))
    z = torch.tensor([near] * 4 + [far] * 4, dtype=torch.float32)
    x, y = (pxy[0] - K[0, 2]) * z / K[0, 0], (pxy[1] - K[1, 2]) * z / K[1, 1]
    Tt = T[:3].t()                                                  # [4, 3]: row j = column j of the pose
    return (((x[:, None] * Tt[0] + y[:, None] * Tt[1]) + z[:, None] * Tt[2]) + Tt[3]).contiguous()


def compute_camera_frustum_corners(depth_map, pose, intrinsics) -> torch.Tensor:
    """Reference: geometry_utils.py:99-129.  Returns f32[8,3] on `pose`'s device."""
    near, far = depth_range(depth_map)
    h, w = depth_map.shape
    out = frustum_corners_from_range(near, far, h, w, pose, intrinsics)
    return out.to(pose.device) if isinstance(pose, torch.Tensor) else out


_PLANE_IDX = tuple(torch.tensor([p[i] for p in _PLANE_DEF]) for i in range(4))


def compute_camera_frustum_planes(frustum_corners) -> torch.Tensor:
    """Reference: geometry_utils.py:163-202.  f32[6,4] rows (a,b,c,d), inside <=> plane.(p,1) <= 0.  The six cross products and
    offsets as one batched call each (bit-equal to six separate torch.linalg.cross / torch.dot calls: tools/host_camera_check.py)."""
    c = _cpu32(frustum_corners)
    a, b, e, f = _PLANE_IDX
    n = torch.linalg.cross(c[a] - c[b], c[e] - c[f])
    p = n * c[:6]                                                   # offset from corner i (sic, :201)
    d = -((p[:, 0] + p[:, 1]) + p[:, 2])
    return torch.cat([n, d[:, None]], dim=1).float()


def compute_frustum_aabb(frustum_corners) -> Tuple[torch.Tensor, torch.Tensor]:
    """Reference: geometry_utils.py:205-215."""
    c = frustum_corners
    return c.min(dim=0).values, c.max(dim=0).values


def make_camera(frustum_corners, w2c, intrinsics, th: float, h: int, w: int) -> L.Camera:
    """Pack the per-frame parameters of csrc/geometry.hip (ovo_camera_t)."""
    cam = L.Camera()
    if frustum_corners is not None:
        c = _cpu32(frustum_corners)
        lo, hi = compute_frustum_aabb(c)
        cam.aabb[:] = torch.cat([lo, hi]).tolist()
        cam.planes[:] = compute_camera_frustum_planes(c).reshape(-1).tolist()
    cam.w2c[:] = _cpu32(w2c).reshape(-1).tolist() if w2c is not None else torch.eye(4).reshape(-1).tolist()
    cam.K[:] = _cpu32(intrinsics).reshape(-1).tolist()
    cam.th, cam.h, cam.w = float(th), int(h), int(w)
    return cam


_last_cam = None
_frame_cams: "dict" = {}          # (near, far, h, w, pose bytes, K bytes) -> 232-byte ovo_camera_t image with th = 0
_FRAME_CAMS_MAX = 256


def _cam_key(near: float, far: float, h: int, w: int, p: torch.Tensor, K: torch.Tensor):
    return (float(near), float(far), int(h), int(w), p.numpy().tobytes(), K.numpy().tobytes())


def _build_cameras(entries, intrinsics: torch.Tensor):
    """ovo_camera_t images (th = 0) of SEVERAL frames in one batch of torch-CPU ops: frustum corners (geometry_utils.py:99-129), their
    AABB (:205-215), the six planes (:163-202) and the inverse pose.  Element for element the arithmetic of the one-frame functions
    above -- every op is elementwise over the batch, `torch.linalg.inv` factorises matrix by matrix -- so the bytes equal theirs
    (tests/test_host_camera.py); what changes is the host time: ~40 small ops per BATCH instead of per frame (0.15 ms each)."""
    K = _cpu32(intrinsics)
    B = len(entries)
    T = torch.stack([e[4] for e in entries])                       # [B, 4, 4]
    h, w = entries[0][2], entries[0][3]
    pxy = _corner_px.get((h, w))
    if pxy is None:
        pxy = _corner_px[(h, w)] = (torch.tensor(_CORNER_X * 2, dtype=torch.float32) * float(w), torch.tensor(_CORNER_Y * 2, dtype=torch.float32) * float(h))
    z = torch.tensor([[e[0]] * 4 + [e[1]] * 4 for e in entries], dtype=torch.float32)         # [B, 8]
    x, y = (pxy[0] - K[0, 2]) * z / K[0, 0], (pxy[1] - K[1, 2]) * z / K[1, 1]
    Tt = T[:, :3].transpose(1, 2)                                   # [B, 4, 3]: row j = column j of the pose
    c = (((x[..., None] * Tt[:, 0:1] + y[..., None] * Tt[:, 1:2]) + z[..., None] * Tt[:, 2:3]) + Tt[:, 3:4]).contiguous()   # [B, 8, 3]
    a, b, e, f = _PLANE_IDX
    n = torch.linalg.cross(c[:, a] - c[:, b], c[:, e] - c[:, f])    # [B, 6, 3]
    p = n * c[:, :6]
    d = -((p[..., 0] + p[..., 1]) + p[..., 2])
    out = np.zeros((B, 58), np.float32)
    out[:, 0:3] = c.min(dim=1).values.numpy()
    out[:, 3:6] = c.max(dim=1).values.numpy()
    out[:, 6:30] = torch.cat([n, d[..., None]], dim=2).reshape(B, 24).numpy()
    out[:, 30:46] = torch.linalg.inv(T).reshape(B, 16).numpy()
    out[:, 46:55] = K.reshape(-1).numpy()
    hw = out.view(np.int32)
    hw[:, 56], hw[:, 57] = h, w
    return [out[i].tobytes() for i in range(B)]


def prepare_frame_cameras(frames, intrinsics) -> None:
    """Compute (and remember) the cameras of several upcoming frames in ONE batch: `frames` = [(depth, pose), ...] with depth maps whose
    range is known without a device sync (numpy, or tagged by `tag_depth_range`).  `frame_camera` then only copies the struct."""
    K = _cpu32(intrinsics)
    todo, seen = [], set()
    for depth, pose in frames:
        if not isinstance(depth, np.ndarray) and getattr(depth, "_ovo_range", None) is None:
            continue
        near, far = depth_range(depth)
        if not far > 0:
            continue
        h, w = depth.shape
        p = _cpu32(pose).contiguous()
        key = _cam_key(near, far, h, w, p, K)
        if key not in _frame_cams and key not in seen:
            seen.add(key)
            todo.append((near, far, int(h), int(w), p, key))
    for hw in {(e[2], e[3]) for e in todo}:
        group = [e for e in todo if (e[2], e[3]) == hw]
        for e, image in zip(group, _build_cameras(group, K)):
            _remember_camera(e[5], image)


def _remember_camera(key, image: bytes) -> None:
    if len(_frame_cams) >= _FRAME_CAMS_MAX:
        for k in list(_frame_cams)[:_FRAME_CAMS_MAX // 2]:
            del _frame_cams[k]
    _frame_cams[key] = image


def frame_camera(near: float, far: float, h: int, w: int, pose: torch.Tensor, intrinsics: torch.Tensor, th: float) -> L.Camera:
    """`make_camera` of a frame's frustum (corners from its depth range).  Cameras are remembered per (range, size, pose, intrinsics):
    the mapper and the tracker ask for the same camera within a keyframe (with their own match thresholds), and a round of keyframes
    is prepared in one batch (`prepare_frame_cameras`); a caller gets its own copy of the struct with its `th`."""
    global _last_cam
    last = _last_cam
    if last is not None and last[2] == (near, far, h, w) and isinstance(pose, torch.Tensor) and isinstance(intrinsics, torch.Tensor) \
            and ((last[0] is pose and pose._version == last[4]) or
                 (pose.shape == last[0].shape and pose.dtype == last[0].dtype and pose.device == last[0].device and torch.equal(last[0], pose))) \
            and ((last[1] is intrinsics and intrinsics._version == last[5]) or
                 (intrinsics.device == last[1].device and intrinsics.dtype == last[1].dtype and torch.equal(last[1], intrinsics))):
        image = last[3]                                            # the same frame asked again (mapper, then tracker): no key to build
    else:
        p, K = _cpu32(pose).contiguous(), _cpu32(intrinsics)
        key = _cam_key(near, far, h, w, p, K)
        image = _frame_cams.get(key)
        if image is None:
            image = _build_cameras([(float(near), float(far), int(h), int(w), p, key)], K)[0]
            _remember_camera(key, image)
        # (the same OBJECT counts as the same pose only at the same version: an in-place update bumps `_version`)
        _last_cam = (pose, intrinsics, (near, far, h, w), image, getattr(pose, "_version", None), getattr(intrinsics, "_version", None))
    cam = L.Camera.from_buffer_copy(image)
    cam.th = float(th)
    return cam


def _count(t: torch.Tensor) -> int:
    return int(t.item())


def compute_frustum_point_ids(pts: torch.Tensor, frustum_corners: torch.Tensor, device: str = "cuda") -> torch.Tensor:
    """Reference: geometry_utils.py:252-276.  Ascending i64 indices of the points inside the frustum."""
    if pts.shape[0] == 0:
        return torch.zeros(0, dtype=torch.int64, device=device)
    pts = L.dev(pts.to(device), torch.float32, "pts")
    n = pts.shape[0]
    cam = make_camera(frustum_corners, None, torch.eye(3), 0.0, 1, 1)
    lib = L.load()
    out = torch.empty(n, dtype=torch.int64, device=pts.device)
    cnt = torch.empty(1, dtype=torch.int64, device=pts.device)
    nb = lib.ovo_compact_workspace_bytes(n)
    ws = L.workspace(nb, pts.device)
    L.check(lib.ovo_frustum_ids(L.ptr(pts), n, cam, L.ptr(out), L.ptr(cnt), L.ptr(ws), nb, L.stream()))
    return out[:_count(cnt)]


def project_3d_points(points_3d: torch.Tensor, intrinsics: torch.Tensor, w2c: torch.Tensor = None) -> torch.Tensor:
    """Reference: geometry_utils.py:26-43.  points_3d f32[N,4] homogeneous -> i32[N,2] pixel (u,v)."""
    pts = L.dev(points_3d, torch.float32, "points_3d")
    n = pts.shape[0]
    cam = make_camera(None, w2c, intrinsics, 0.0, 1, 1)
    out = torch.empty((n, 2), dtype=torch.int32, device=pts.device)
    L.check(L.load().ovo_project_points(L.ptr(pts), n, pts.shape[1], cam, L.ptr(out), L.stream()))
    return out


def match_3d_points_to_2d_pixels(depth: torch.Tensor, w2c: torch.Tensor, points_3d: torch.Tensor,
                                 intrinsics: torch.Tensor, th_dist: float) -> Tuple[torch.Tensor, torch.Tensor]:
    """Reference: geometry_utils.py:46-89.  Returns (i64[M] indices into points_3d, i32[M,2] (u,v))."""
    depth = L.dev(depth, torch.float32, "depth")
    pts = L.dev(points_3d, torch.float32, "points_3d")
    n = pts.shape[0]
    h, w = depth.shape
    cam = make_camera(None, w2c, intrinsics, th_dist, h, w)
    lib = L.load()
    idx = torch.empty(n, dtype=torch.int64, device=pts.device)
    uv = torch.empty((n, 2), dtype=torch.int32, device=pts.device)
    cnt = torch.empty(1, dtype=torch.int64, device=pts.device)
    nb = lib.ovo_compact_workspace_bytes(max(n, 1))
    ws = L.workspace(nb, pts.device)
    stride = pts.shape[1] if n else 3
    L.check(lib.ovo_match_points(L.ptr(pts), n, stride, cam, L.ptr(depth), L.ptr(idx), L.ptr(uv), L.ptr(cnt),
                                 L.ptr(ws), nb, L.stream()))
    m = _count(cnt)
    return idx[:m], uv[:m]


def depth_filter(depth: torch.Tensor, k_size: int = 7, sigma: float = 2.5, th: float = 0.05) -> torch.Tensor:
    """Reference: geometry_utils.py:92-96.  Pixels whose high-pass response exceeds `th` become -1."""
    depth = L.dev(depth, torch.float32, "depth")
    out = torch.empty_like(depth)
    h, w = depth.shape
    L.check(L.load().ovo_depth_filter(L.ptr(depth), h, w, int(k_size), float(sigma), float(th), L.ptr(out), L.stream()))
    return out
