"""-m gpu: HIP geometry / mapping / tracking path vs the oracle and the reference's golden vectors.

Integer and index outputs (and fp32 coordinates produced by the FMA chain) are compared bit-exactly.
"""
import numpy as np
import pytest
import torch

from conftest import golden, unpack

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _t(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t if dtype is None else t.to(dtype)).to(DEV)


# ------------------------------------------------------------------ golden (reference outputs)
@pytest.mark.parametrize("t", [1, 2])
def test_frustum_project_match_golden(t):
    from ovo_amd.utils import geometry_utils as G
    d = golden(f"geometry_t{t}")
    pts = _t(d["pts"])
    corners = G.compute_camera_frustum_corners(d["depth"], torch.from_numpy(d["c2w"]), torch.from_numpy(d["K"]))
    assert np.array_equal(corners.numpy(), d["corners"])
    ids = G.compute_frustum_point_ids(pts, torch.from_numpy(d["corners"]), device=DEV)
    assert ids.dtype == torch.int64 and np.array_equal(ids.cpu().numpy(), d["frustum_ids"])
    fp = pts[ids]
    hom = torch.hstack([fp, torch.ones((fp.shape[0], 1), device=DEV)]).contiguous()
    uv = G.project_3d_points(hom, torch.from_numpy(d["K"]), torch.from_numpy(d["w2c"]))
    assert uv.dtype == torch.int32 and np.array_equal(uv.cpu().numpy(), d["project_uv"])
    mi, muv = G.match_3d_points_to_2d_pixels(_t(d["depth"]), torch.from_numpy(d["w2c"]), fp.contiguous(),
                                             torch.from_numpy(d["K"]), float(d["th"]))
    assert np.array_equal(mi.cpu().numpy(), d["match_idx"]) and np.array_equal(muv.cpu().numpy(), d["match_uv"])
    # homogeneous input gives the same answer
    mi4, muv4 = G.match_3d_points_to_2d_pixels(_t(d["depth"]), torch.from_numpy(d["w2c"]), hom, torch.from_numpy(d["K"]), float(d["th"]))
    assert torch.equal(mi, mi4) and torch.equal(muv, muv4)


def test_vanilla_mapper_golden():
    from ovo_amd.slam.vanilla_mapper import VanillaMapper
    d = golden("vanilla_mapper")
    vm = VanillaMapper({"device": DEV, "mapping": {"k_pooling": 3}}, torch.from_numpy(d["K"]).to(DEV))
    for i in range(3):
        fd = [i, d[f"rgb{i}"], d[f"depth{i}"], d[f"c2w{i}"]]
        vm.track_camera(fd)
        vm.map(fd, vm.get_c2w(i))
        assert vm.pcd.shape[0] == int(d[f"n{i}"])
    assert np.array_equal(vm.pcd.cpu().numpy(), d["pcd"])
    assert np.array_equal(vm.pcd_ids.cpu().numpy(), d["pcd_ids"])
    assert np.array_equal(vm.pcd_obj_ids.cpu().numpy(), d["pcd_obj_ids"])
    assert np.array_equal(vm.pcd_colors.cpu().numpy(), d["pcd_colors"])
    md = vm.get_map_dict()
    assert md["xyz"].shape == (vm.max_id, 3) and md["obj_ids"].shape == (vm.max_id, 1) and md["ids"].dtype == torch.int32


class _FixedMasks:
    def __init__(self):
        self.next = None

    def get_masks(self, image, frame_id):
        seg, masks = self.next
        return torch.from_numpy(seg).to(DEV), torch.from_numpy(masks).to(DEV)


class _NoClip:
    clip_dim = 16


@pytest.mark.parametrize("tag,filt", [("nofilter", False), ("filter", True), ("ratio", True)])
def test_tracking_golden(tag, filt):
    from ovo_amd.entities.ovo import OVO
    from ovo_amd.slam.vanilla_mapper import VanillaMapper
    d = golden(f"tracking_{tag}")
    w = int(d["mask_w"])
    ratio = tuple(d["ratio"].tolist())
    ratio = (ratio[0], ratio[1], int(ratio[2])) if ratio else ()
    K = torch.from_numpy(d["K"]).to(DEV)
    cfg = {"match_distance_th": 0.05, "track_th": int(d["track_th"]), "depth_filter": filt, "log": False,
           "debug_info": True, "clip": {"k_top_views": int(d["n_top_views"]), "fusion": "avg_pooling"}, "sam": {}}
    mg = _FixedMasks()
    ovo = OVO(cfg, None, None, K, device=DEV, clip_generator=_NoClip(), mask_generator=mg)
    vm = VanillaMapper({"device": DEV, "mapping": {}}, K)
    for i in range(4):
        fd = [i, d[f"rgb{i}"], d[f"depth{i}"], d[f"c2w{i}"]]
        vm.track_camera(fd)
        vm.map(fd, vm.get_c2w(i))
        assert vm.pcd.shape[0] == int(d[f"pcd_n{i}"])
        assert np.array_equal(vm.get_map()[2].cpu().numpy(), d[f"ins_before{i}"])
        masks = unpack(d[f"masks{i}"], w)
        mg.next = (d[f"seg{i}"], masks)
        updated = ovo.detect_and_track_objects([i, d[f"rgb{i}"], d[f"depth{i}"], ratio], vm.get_map(), vm.get_c2w(i))
        assert updated.dtype == torch.int32 and np.array_equal(updated.cpu().numpy(), d[f"updated{i}"])
        vm.update_pcd_obj_ids(updated)
        matched, fused, _, kf = ovo.keyframes_queue[-1]
        assert kf == i and matched == d[f"matched_ins_ids{i}"].tolist()
        assert np.array_equal(fused.cpu().numpy(), unpack(d[f"bmaps{i}"], w))
        assert ovo.next_ins_id == int(d[f"next_ins_id{i}"])
    assert sorted(ovo.objects) == d["obj_ids"].tolist()
    for j, o in ovo.objects.items():
        assert o.kfs_ids == d[f"obj{j}_kfs"].tolist()
        assert sorted(o.top_kf) == [tuple(r) for r in d[f"obj{j}_topkf"].tolist()]
        assert np.asarray(o.points_ids).reshape(-1).tolist() == d[f"obj{j}_points"].reshape(-1).tolist()


@pytest.mark.parametrize("tag,filt", [("nofilter", False), ("filter", True), ("ratio", True)])
def test_tracking_golden_device_decisions(tag, filt):
    """The same reference goldens through the chain that takes the decisions of ovo.py:255-324 on the device (`ovo_track_step`:
    no debug exports, so the per-instance point-id lists are not produced)."""
    from ovo_amd.entities.ovo import OVO
    from ovo_amd.slam.vanilla_mapper import VanillaMapper
    d = golden(f"tracking_{tag}")
    w = int(d["mask_w"])
    ratio = tuple(d["ratio"].tolist())
    ratio = (ratio[0], ratio[1], int(ratio[2])) if ratio else ()
    K = torch.from_numpy(d["K"]).to(DEV)
    cfg = {"match_distance_th": 0.05, "track_th": int(d["track_th"]), "depth_filter": filt, "log": False,
           "debug_info": False, "clip": {"k_top_views": int(d["n_top_views"]), "fusion": "avg_pooling"}, "sam": {}}
    mg = _FixedMasks()
    ovo = OVO(cfg, None, None, K, device=DEV, clip_generator=_NoClip(), mask_generator=mg)
    vm = VanillaMapper({"device": DEV, "mapping": {}}, K)
    for i in range(4):
        fd = [i, d[f"rgb{i}"], d[f"depth{i}"], d[f"c2w{i}"]]
        vm.track_camera(fd)
        vm.map(fd, vm.get_c2w(i))
        assert vm.pcd.shape[0] == int(d[f"pcd_n{i}"])
        masks = unpack(d[f"masks{i}"], w)
        mg.next = (d[f"seg{i}"], masks)
        updated = ovo.detect_and_track_objects([i, d[f"rgb{i}"], d[f"depth{i}"], ratio], vm.get_map(), vm.get_c2w(i))
        assert updated.dtype == torch.int32 and np.array_equal(updated.cpu().numpy(), d[f"updated{i}"])
        vm.update_pcd_obj_ids(updated)
        matched, fused, _, kf = ovo.keyframes_queue[-1]
        assert kf == i and matched == d[f"matched_ins_ids{i}"].tolist()
        assert np.array_equal(fused.cpu().numpy(), unpack(d[f"bmaps{i}"], w))
        assert ovo.next_ins_id == int(d[f"next_ins_id{i}"])
    assert ovo._track_ring is not None and ovo._track_ring.seq == 4      # the device-decision chain really ran
    assert sorted(ovo.objects) == d["obj_ids"].tolist()
    for j, o in ovo.objects.items():
        assert o.kfs_ids == d[f"obj{j}_kfs"].tolist()
        assert sorted(o.top_kf) == [tuple(r) for r in d[f"obj{j}_topkf"].tolist()]


def _run_keyframes(queued, n_frames=6, top_k=2, mask_grid=(3, 4), n_blobs=5, track_th=30):
    """n keyframes of the synthetic stream through mapper + tracker; `queued`: all map / tracking chains are launched back to back
    (sizes and instance ids device-resident) and finished afterwards -- "chain": all of them in ONE launch of the persistent round
    kernel (`ovo_round_chain`) -- else one keyframe at a time with host decisions."""
    from ovo_amd import synthetic as syn
    from ovo_amd.entities.ovo import OVO
    from ovo_amd.slam.vanilla_mapper import VanillaMapper
    scale = 0.5
    h, w = syn.scannet_depth_hw(scale)
    K = torch.from_numpy(syn.scannet_intrinsics(scale)).to(DEV)

    class Masks:
        def get_masks(self, image, frame_id):
            m = syn.make_masks(h, w, grid=mask_grid, n_blobs=n_blobs, seed=frame_id)
            return torch.from_numpy(syn.masks_to_segmap(m)).to(DEV), torch.from_numpy(m).to(DEV)

    cfg = {"match_distance_th": 0.05, "track_th": track_th, "depth_filter": True, "log": False, "debug_info": False, "host_decisions": not queued,
           "clip": {"k_top_views": top_k, "fusion": "avg_pooling"}, "sam": {}}
    ovo = OVO(cfg, None, None, K, device=DEV, clip_generator=_NoClip(), mask_generator=Masks())
    vm = VanillaMapper({"device": DEV, "mapping": {}}, K)
    frames = [syn.frame(t, scale=scale, seed=11) for t in range(n_frames)]
    if queued in ("chain", "merged"):
        from ovo_amd.entities.round_chain import RoundLauncher
        vm.reserve(n_frames * h * w)
        maps, tracks, pend = [], [], []
        for fid, rgb, depth, c2w in frames:
            fd = [fid, rgb, depth, c2w]
            vm.track_camera(fd)
            maps.append(vm.map_launch(fd, vm._c2w_host[fid], defer=True))
            pend.append(ovo.detect_and_track_launch([fid, rgb, depth, ()], vm, vm._c2w_host[fid], defer=True))
            tracks.append(pend[-1]["step"])
        launcher = RoundLauncher(DEV, workgroups=48)
        if queued == "merged":                                       # ovo_keyframe_step: 7 launches per keyframe, independent passes share launches
            launcher.enabled = False
        launcher.launch(maps, tracks)
        vm.launched()
        assert (launcher.launches, launcher.fallbacks) == ((0, 1) if queued == "merged" else (1, 0))
        for p in pend:
            ovo.detect_and_track_finish(p)
    elif queued:
        vm.reserve(n_frames * h * w)
        pend = []
        for fid, rgb, depth, c2w in frames:
            fd = [fid, rgb, depth, c2w]
            vm.track_camera(fd)
            vm.map_launch(fd, vm._c2w_host[fid])
            pend.append(ovo.detect_and_track_launch([fid, rgb, depth, ()], vm, vm._c2w_host[fid]))
        assert vm._ring.seq == n_frames and len(ovo._track_pending) == n_frames            # everything queued before anything is finished
        for p in pend:
            ovo.detect_and_track_finish(p)
    else:
        for fid, rgb, depth, c2w in frames:
            fd = [fid, rgb, depth, c2w]
            vm.track_camera(fd)
            vm.map(fd, vm.get_c2w(fid))
            upd = ovo.detect_and_track_objects([fid, rgb, depth, ()], vm.get_map(), vm.get_c2w(fid))
            vm.update_pcd_obj_ids(upd)
        assert ovo._track_ring is None                              # host decisions
    return {"pcd": vm.pcd.cpu(), "ids": vm.pcd_ids.cpu(), "ins": vm.pcd_obj_ids.cpu(), "rgb": vm.pcd_colors.cpu(), "max_id": vm.max_id,
            "next": ovo.next_ins_id, "objects": {i: (list(o.kfs_ids), sorted(o.top_kf), o.to_update) for i, o in ovo.objects.items()},
            "queue": [(m, b.cpu(), kf) for m, b, _, kf in ovo.keyframes_queue]}


@pytest.mark.parametrize("mode", [True, "merged", "chain"])
def test_queued_keyframe_chains_equal_one_by_one_host_decisions(mode):
    """Six keyframes queued back to back on the device (map size, point ids, instance ids resident; one result block each) -- as
    ~13 launches per keyframe, as 7 (independent passes merged, `ovo_keyframe_step`), or all in ONE persistent launch with grid barriers -- give the map, the instance list, the heaps and
    the fused masks of the keyframe-at-a-time run with host decisions, bit for bit."""
    if mode == "chain":
        from ovo_amd import _lib as L
        if L.load().ovo_round_chain_params_bytes() == 0:
            pytest.skip("k_round_chain is not in a production build (python -m ovo_amd.build --force --experimental)")
    a, b = _run_keyframes(mode), _run_keyframes(False)
    assert a["max_id"] == b["max_id"] and a["next"] == b["next"] and a["next"] > 5
    for k in ("pcd", "ids", "ins", "rgb"):
        assert torch.equal(a[k], b[k]), k
    assert (a["ins"] >= 0).sum() > 1000
    assert a["objects"] == b["objects"]
    assert any(len(top) > 0 for _, top, _ in a["objects"].values())
    assert len(a["queue"]) == len(b["queue"]) == 6
    for (m1, b1, k1), (m2, b2, k2) in zip(a["queue"], b["queue"]):
        assert m1 == m2 and k1 == k2 and torch.equal(b1, b2)


@pytest.mark.parametrize("mode", [True, "merged"])
def test_many_masks_per_keyframe_device_vs_host_decisions(mode):
    """168 masks per keyframe (a 9 x 12 grid + 60 blobs that overlap it), low track threshold: several masks vote for the same instance, their
    indices more than 64 apart -- the ballot walk of the mask fusion takes several steps, the decision scan several masks per thread -- and the
    device decisions, fused masks and heaps still equal the host-decision run bit for bit."""
    kw = dict(n_frames=4, mask_grid=(9, 12), n_blobs=60, track_th=6)
    a, b = _run_keyframes(mode, **kw), _run_keyframes(False, **kw)
    assert a["max_id"] == b["max_id"] and a["next"] == b["next"] and a["next"] > 60
    for k in ("pcd", "ids", "ins", "rgb"):
        assert torch.equal(a[k], b[k]), k
    assert a["objects"] == b["objects"]
    fused = 0
    for (m1, b1, k1), (m2, b2, k2) in zip(a["queue"], b["queue"]):
        assert m1 == m2 and k1 == k2 and torch.equal(b1, b2)
        fused += len(m1) - len(set(m1)) if isinstance(m1, (list, tuple)) else 0
    n_masks = [len(m) for m, _, _ in a["queue"]]
    assert max(n_masks) > 64, n_masks


# ------------------------------------------------------------------ oracle at full size
def _scene(n_points, scale=1.0, t=2, seed=5):
    from ovo_amd import synthetic as syn
    h, w = syn.scannet_depth_hw(scale)
    K = syn.scannet_intrinsics(scale)
    c2w = syn.pose(t)
    depth = syn.render_depth(c2w, K, h, w, seed=seed)
    pts = syn.padded_map(n_points, frames=3, scale=scale, seed=seed)
    return pts, depth, c2w, K


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 100_003, 1_000_000])
def test_frustum_and_match_vs_oracle_sizes(n):
    from oracle import geometry as OG
    from ovo_amd.utils import geometry_utils as G
    pts, depth, c2w, K = _scene(max(n, 1))
    pts = pts[:n]
    corners = OG.frustum_corners(depth, c2w, K)
    ids = G.compute_frustum_point_ids(_t(pts).reshape(-1, 3), torch.from_numpy(corners), device=DEV)
    ref = OG.frustum_point_ids(pts, corners)
    assert np.array_equal(ids.cpu().numpy(), ref)
    if n == 0:
        return
    w2c = torch.linalg.inv(torch.from_numpy(c2w))
    fp = np.ascontiguousarray(pts[ref])
    mi, muv = G.match_3d_points_to_2d_pixels(_t(depth), w2c, _t(fp).reshape(-1, 3), torch.from_numpy(K), 0.05)
    ri, ruv = OG.match(depth, w2c.numpy(), fp, K, 0.05)
    assert np.array_equal(mi.cpu().numpy(), ri) and np.array_equal(muv.cpu().numpy(), ruv)
    if n >= 100_000:
        assert ri.shape[0] > 1000          # the scene really exercises the matcher
        assert (np.diff(mi.cpu().numpy()) > 0).all()      # ascending, unique


def test_depth_filter_vs_oracle():
    from oracle import geometry as OG
    from ovo_amd.utils import geometry_utils as G
    _, depth, _, _ = _scene(1)
    depth[100:140, 200:260] += 0.4            # a step edge so the filter fires
    out = G.depth_filter(_t(depth)).cpu().numpy()
    ref = OG.depth_filter(depth)
    diff = out != ref
    # fp32 vs fp64 accumulation: only pixels within 1e-5 of the threshold may flip
    assert diff.mean() < 1e-4
    assert (ref == -1).sum() > 100 and (out == -1).sum() > 100
    same = ~diff
    assert np.array_equal(out[same], ref[same])


def test_track_project_matches_unfused_oracle_full_size():
    """Fused cull+project+match+seg-lookup+vote pass == oracle composition, 1M-point map, with colour/depth ratio."""
    from oracle import geometry as OG
    from ovo_amd import _lib as L, synthetic as syn
    from ovo_amd.utils import geometry_utils as G
    n = 1_000_000
    pts, depth, c2w, K = _scene(n)
    h, w = depth.shape
    H, W = 480, 640
    masks = syn.make_masks(H, W, seed=9)
    seg = syn.masks_to_segmap(masks)
    rng = np.random.default_rng(3)
    ins = np.where(rng.random(n) < 0.3, -1, rng.integers(0, 2000, n)).astype(np.int32)
    ratio = (1.0, 1.0, 12)
    corners = OG.frustum_corners(depth, c2w, K)
    w2c = torch.linalg.inv(torch.from_numpy(c2w))
    cam = G.make_camera(torch.from_numpy(corners), w2c, torch.from_numpy(K), 0.05, h, w)
    n_masks, cols = masks.shape[0], 2001
    point_seg = torch.empty(n, dtype=torch.int16, device=DEV)
    hist = torch.empty((n_masks, cols), dtype=torch.int32, device=DEV)
    counters = torch.empty(2, dtype=torch.int64, device=DEV)
    lib = L.load()
    d_pts, d_ins, d_depth, d_seg = _t(pts), _t(ins), _t(depth), _t(seg)     # keep alive: raw pointers cross the ABI
    L.check(lib.ovo_track_project(L.ptr(d_pts), L.ptr(d_ins), n, cam, L.ptr(d_depth), L.ptr(d_seg), H, W,
                                  L.Ratio(1, ratio[0], ratio[1], ratio[2]), L.ptr(point_seg), L.ptr(hist), n_masks, cols,
                                  L.ptr(counters), L.stream()))
    stats = torch.empty((n_masks, 4), dtype=torch.int32, device=DEV)
    L.check(lib.ovo_vote_stats(L.ptr(hist), n_masks, cols, L.ptr(d_seg), seg.size, L.ptr(stats), L.stream()))
    # oracle composition
    fids = OG.frustum_point_ids(pts, corners)
    midx, uv = OG.match(depth, w2c.numpy(), pts[fids], K, 0.05)
    uv = uv + 12
    ref_seg = np.full(n, -2, np.int16)
    ref_seg[fids[midx]] = seg[uv[:, 1], uv[:, 0]]
    assert counters.tolist() == [fids.shape[0], midx.shape[0]]
    assert np.array_equal(point_seg.cpu().numpy(), ref_seg)
    ref_hist = np.zeros((n_masks, cols), np.int64)
    sel = ref_seg >= 0
    np.add.at(ref_hist, (ref_seg[sel].astype(np.int64), ins[sel].astype(np.int64) + 1), 1)
    assert np.array_equal(hist.cpu().numpy(), ref_hist)
    st = stats.cpu().numpy()
    assert np.array_equal(st[:, 0], ref_hist.sum(1)) and np.array_equal(st[:, 1], ref_hist[:, 1:].sum(1))
    assert np.array_equal(st[:, 3], np.bincount(seg[seg >= 0], minlength=n_masks))
    for m in range(n_masks):
        row = ref_hist[m, 1:]
        exp = -1 if row.max() == 0 else int(np.flatnonzero(row == row.max())[0])
        assert st[m, 2] == exp
    # write half: idempotent, only touches free points of targeted masks
    target = np.where(np.arange(n_masks) % 2 == 0, 5000 + np.arange(n_masks), -1).astype(np.int32)
    out = torch.empty(n, dtype=torch.int32, device=DEV)
    cnt = torch.empty(1, dtype=torch.int64, device=DEV)
    d_target = _t(target)
    L.check(lib.ovo_assign_instances(L.ptr(d_ins), L.ptr(point_seg), n, L.ptr(d_target), n_masks, L.ptr(out), L.ptr(cnt), L.stream()))
    exp = ins.copy()
    hit = (ref_seg >= 0) & (ins == -1)
    tg = target[np.clip(ref_seg, 0, None)]
    exp[hit & (tg > -1)] = tg[hit & (tg > -1)]
    assert np.array_equal(out.cpu().numpy(), exp) and int(cnt) == int((exp != ins).sum())
    out2 = torch.empty_like(out)
    L.check(lib.ovo_assign_instances(L.ptr(out), L.ptr(point_seg), n, L.ptr(d_target), n_masks, L.ptr(out2), None, L.stream()))
    assert torch.equal(out, out2)
