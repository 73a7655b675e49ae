"""CPU: the batched frustum set-up on the host (a1: geometry_utils.frustum_corners_from_range / compute_camera_frustum_planes) is bit-equal
to the op-by-op form that mirrors the reference line by line (geometry_utils.py:99-129, 163-202) on random cameras -- the planes feed
inside / outside tests whose results are compared bit-exactly with the reference's, so not one ulp may move."""
import numpy as np
import torch


def _corners_ref(G, near, far, h, w, T, K):
    px = torch.tensor(G._CORNER_X * 2, dtype=torch.float32) * float(w)
    py = torch.tensor(G._CORNER_Y * 2, dtype=torch.float32) * float(h)
    z = torch.tensor([near] * 4 + [far] * 4, dtype=torch.float32)
    cam = torch.stack([(px - K[0, 2]) * z / K[0, 0], (py - K[1, 2]) * z / K[1, 1], z, torch.ones(8)], dim=1)
    return torch.einsum("ij,mj->mi", T, cam)[:, :3].contiguous()


def _planes_ref(G, c):
    n = torch.stack([torch.linalg.cross(c[a] - c[b], c[e] - c[f]) for a, b, e, f in G._PLANE_DEF])
    d = torch.stack([-torch.dot(n[i], c[i]) for i in range(6)])
    return torch.cat([n, d[:, None]], dim=1).float()


def test_batched_frustum_setup_is_bit_equal_to_the_op_by_op_form():
    from ovo_amd.utils import geometry_utils as G
    rng = np.random.default_rng(0)
    for _ in range(400):
        R = np.linalg.qr(rng.standard_normal((3, 3)))[0].astype(np.float32)
        T = np.eye(4, dtype=np.float32)
        T[:3, :3], T[:3, 3] = R, (rng.standard_normal(3) * 3).astype(np.float32)
        K = torch.tensor([[500 + rng.random() * 100, 0, 320 + rng.random() * 5], [0, 500 + rng.random() * 100, 240 + rng.random() * 5], [0, 0, 1]],
                         dtype=torch.float32)
        near, far = float(np.float32(0.3 + rng.random())), float(np.float32(2 + rng.random() * 5))
        h, w = (480, 640) if rng.random() < 0.5 else (456, 616)
        cr = _corners_ref(G, near, far, h, w, torch.from_numpy(T), K)
        assert torch.equal(cr, G.frustum_corners_from_range(near, far, h, w, torch.from_numpy(T), K))
        assert torch.equal(_planes_ref(G, cr), G.compute_camera_frustum_planes(cr))


def test_frame_camera_is_shared_across_thresholds():
    from ovo_amd.utils import geometry_utils as G
    K = torch.tensor([[500., 0, 320], [0, 500, 240], [0, 0, 1]])
    P = torch.eye(4)
    a = G.frame_camera(0.5, 3.0, 480, 640, P, K, 0.03)
    b = G.frame_camera(0.5, 3.0, 480, 640, P, K, 0.05)
    assert abs(a.th - 0.03) < 1e-7 and abs(b.th - 0.05) < 1e-7 and list(a.planes) == list(b.planes) and list(a.w2c) == list(b.w2c)
    direct = G.make_camera(G.frustum_corners_from_range(0.5, 3.0, 480, 640, P, K), torch.linalg.inv(P), K, 0.05, 480, 640)
    assert bytes(direct) == bytes(b)


def test_batched_cameras_equal_one_by_one():
    """`prepare_frame_cameras` (one batch of torch-CPU ops for a round of keyframes) produces the bytes the per-frame path does."""
    from ovo_amd.utils import geometry_utils as G
    rng = np.random.default_rng(1)
    K = torch.tensor([[577.6, 0, 318.9], [0, 578.7, 242.7], [0, 0, 1]], dtype=torch.float32)
    frames, singles = [], []
    for _ in range(24):
        R = np.linalg.qr(rng.standard_normal((3, 3)))[0].astype(np.float32)
        T = np.eye(4, dtype=np.float32)
        T[:3, :3], T[:3, 3] = R, (rng.standard_normal(3) * 3).astype(np.float32)
        depth = (0.3 + 4 * rng.random((456, 616))).astype(np.float32)
        near, far = G.depth_range(depth)
        corners = G.frustum_corners_from_range(near, far, 456, 616, torch.from_numpy(T), K)
        singles.append(bytes(G.make_camera(corners, torch.linalg.inv(torch.from_numpy(T)), K, 0.05, 456, 616)))
        frames.append((depth, T))
    G._frame_cams.clear()
    G.prepare_frame_cameras(frames, K)
    assert len(G._frame_cams) == 24
    for (depth, T), ref in zip(frames, singles):
        near, far = G.depth_range(depth)
        assert bytes(G.frame_camera(near, far, 456, 616, torch.from_numpy(T), K, 0.05)) == ref
    G._frame_cams.clear()
    for (depth, T), ref in zip(frames, singles):               # and the unbatched look-up
        near, far = G.depth_range(depth)
        assert bytes(G.frame_camera(near, far, 456, 616, torch.from_numpy(T), K, 0.05)) == ref
