"""The HBM-bound geometry passes at BASELINE.json's map sizes (1 M / 5 M / 10 M points): time per launch (hipEvents) and algorithmic
GB/s against SURVEY.md section 8d's per-point byte counts.  Run it under rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc
WRITE_SIZE for the kernel-side view (profiles/r02_geom_*).    python tools/geom_bench.py [N ...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ovo_amd import _lib as L, synthetic as syn
from ovo_amd.utils import geometry_utils as G
dev = torch.device("cuda", 0)
lib = L.load()
sizes = [int(v) for v in sys.argv[1:]] or [1_000_000, 5_000_000, 10_000_000]
K = syn.scannet_intrinsics(1.0)
fid, rgb, depth_np, c2w = syn.frame(3, scale=1.0, seed=0)
h, w = depth_np.shape
depth = torch.from_numpy(depth_np).to(dev)
masks = syn.make_masks(h + 24, w + 24, grid=(4, 6), n_blobs=8, seed=3)
seg = torch.from_numpy(syn.masks_to_segmap(masks)).to(dev)
pose = torch.from_numpy(c2w).float()
Kt = torch.from_numpy(K).float()
near, far = G.depth_range(depth_np)
corners = G.frustum_corners_from_range(near, far, h, w, pose, Kt)
cam = G.make_camera(corners, torch.linalg.inv(pose), Kt, 0.05, h, w)
ITERS = int(os.environ.get("ITERS", "10"))

def timed(fn):
    for _ in range(10): fn()                      # (2 warm-up launches left clock ramp / first-touch cost in the first row of a fresh process: 130 vs 24 us)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(ITERS): fn()
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / ITERS     # us

print("%-10s %-22s %10s %12s   %s" % ("points", "pass", "us", "GB/s (alg)", "algorithmic bytes"))
for n in sizes:
    pts = torch.from_numpy(syn.padded_map(n, frames=4, scale=1.0, seed=0)).to(dev)
    ins = torch.full((n,), -1, dtype=torch.int32, device=dev)
    ins[::3] = torch.arange(0, (n + 2) // 3, dtype=torch.int32, device=dev) % 200          # a third of the map already belongs to 200 instances
    n_masks, cols = int(masks.shape[0]), 201
    point_seg = torch.empty(n, dtype=torch.int16, device=dev)
    hist = torch.empty((n_masks, cols), dtype=torch.int32, device=dev)
    small = torch.empty(n_masks * 4 + 4, dtype=torch.int32, device=dev)
    r = L.Ratio(1, 1.0, 1.0, 12)
    def track():
        L.check(lib.ovo_track_project(L.ptr(pts), L.ptr(ins), n, cam, L.ptr(depth), L.ptr(seg), seg.shape[0], seg.shape[1], r, L.ptr(point_seg), L.ptr(hist),
                                      n_masks, cols, small[n_masks * 4:].data_ptr(), L.stream()))
    t = timed(track)
    n_f = int(small[n_masks * 4:].view(torch.int64)[0].item())             # points inside the frustum (they also gather depth + seg: 8 B each)
    b = 14.0 * n + 8.0 * n_f
    print("%-10d %-22s %10.1f %12.1f   12 B xyz + 2 B mask id per point, + 8 B depth/seg gather per in-frustum point (%d)" % (n, "track_project", t, b / t / 1e3, n_f))
    target = torch.arange(n_masks, dtype=torch.int32, device=dev)
    updated = torch.empty_like(ins)
    def assign():
        L.check(lib.ovo_assign_instances(L.ptr(ins), L.ptr(point_seg), n, L.ptr(target), n_masks, L.ptr(updated), None, L.stream()))
    t = timed(assign)
    print("%-10d %-22s %10.1f %12.1f   4 + 2 B read, 4 B written per point" % (n, "assign_instances", t, 10.0 * n / t / 1e3))
    out = torch.empty(n, dtype=torch.int64, device=dev)
    cnt = torch.empty(1, dtype=torch.int64, device=dev)
    nb = lib.ovo_compact_workspace_bytes(n)
    ws = L.workspace(nb, dev)
    def frustum():
        L.check(lib.ovo_frustum_ids(L.ptr(pts), n, cam, L.ptr(out), L.ptr(cnt), L.ptr(ws), nb, L.stream()))
    t = timed(frustum)
    m = int(cnt.item())
    print("%-10d %-22s %10.1f %12.1f   2 x 12 B xyz (flag + emit passes) + 8 B per emitted index (%d)" % (n, "frustum_ids", t, (24.0 * n + 8.0 * m) / t / 1e3, m))
    idx = torch.empty(n, dtype=torch.int64, device=dev)
    uv = torch.empty((n, 2), dtype=torch.int32, device=dev)
    def match():
        L.check(lib.ovo_match_points(L.ptr(pts), n, 3, cam, L.ptr(depth), L.ptr(idx), L.ptr(uv), L.ptr(cnt), L.ptr(ws), nb, L.stream()))
    t = timed(match)
    m = int(cnt.item())
    print("%-10d %-22s %10.1f %12.1f   2 x 12 B xyz + 16 B per matched point (%d)" % (n, "match_points", t, (24.0 * n + 16.0 * m) / t / 1e3, m))
    explained = torch.empty((h, w), dtype=torch.uint8, device=dev)
    def expl():
        L.check(lib.ovo_map_explained(L.ptr(pts), n, cam, L.ptr(depth), L.ptr(explained), L.stream()))
    t = timed(expl)
    print("%-10d %-22s %10.1f %12.1f   12 B xyz per point" % (n, "map_explained", t, 12.0 * n / t / 1e3))
    del pts, ins, point_seg, out, idx, uv
    torch.cuda.empty_cache()
