"""-m gpu: loop-closure semantic update (f3, ovo.py:366-424) -- OVO.update_map vs the oracle's literal restatement."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


class _NoClip:
    clip_dim = 32


def _blob(rng, centre, n, size=0.3):
    return (np.asarray(centre, np.float32) + rng.uniform(-size, size, (n, 3))).astype(np.float32)


def _scene(seed=0):
    """8 instances: (0,1) duplicates -> merge by p_dist > 0.5; (2,3) similar descriptors, half overlapping -> merge by the
    cos > 0.9 & p_dist > 0.2 clause; 4 close to 0 but a different descriptor; 5 same descriptor as 0 but 0.6 m away (no close
    points); 6 far away; 7 has lost all its points."""
    rng = np.random.default_rng(seed)
    base = rng.standard_normal((8, 32)).astype(np.float32)
    feats = {i: base[i] / np.linalg.norm(base[i]) for i in range(8)}
    feats[1] = feats[0] + 0.15 * feats[1]
    feats[3] = feats[2] + 0.05 * feats[3]
    feats[5] = feats[0] + 0.02 * feats[5]
    a = _blob(rng, (0, 0, 0), 3000)
    parts = {0: a, 1: a[:2500] + rng.normal(0, 0.01, (2500, 3)).astype(np.float32),
             2: _blob(rng, (3, 0, 0), 2000), 3: np.concatenate([_blob(rng, (3.1, 0, 0), 700), _blob(rng, (3.9, 0.5, 0), 1300, 0.2)]),
             4: _blob(rng, (0.1, 0.1, 0), 1500), 5: _blob(rng, (0.0, 1.2, 0.0), 1800, 0.25), 6: _blob(rng, (9, 9, 2), 1000)}
    xyz = np.concatenate([parts[i] for i in parts] + [_blob(rng, (5, 5, 5), 500)])
    ins = np.concatenate([np.full(len(parts[i]), i, np.int32) for i in parts] + [np.full(500, -1, np.int32)])
    perm = rng.permutation(len(xyz))
    return xyz[perm], ins[perm], feats


def test_update_map_vs_oracle():
    from oracle import semantic as OS
    from ovo_amd.entities.descriptor_bank import KeyframeView
    from ovo_amd.entities.instance3d import Instance3D
    from ovo_amd.entities.ovo import OVO
    xyz, ins, feats = _scene()
    K = torch.eye(3)
    cfg = {"match_distance_th": 0.05, "track_th": 40, "clip": {"k_top_views": 10, "fusion": "avg_pooling"}, "sam": {"precomputed": True}}
    ovo = OVO(cfg, None, "scene", K.to(DEV), device=DEV, clip_generator=_NoClip(), mask_generator=object())
    ids = list(range(8))
    # every instance was seen in two keyframes (kf = id and kf = id + 8); descriptors: its feature and a scaled copy
    ovo.keyframes["frame_id"] = list(range(0, 160, 10))
    for kf in range(16):
        i = kf % 8
        row = ovo.bank.append(torch.from_numpy(feats[i] * (1.0 if kf < 8 else 0.5))[None].to(DEV))[0]
        ovo.keyframes["ins_descriptors"][kf] = KeyframeView(ovo.bank, {i: row})
    for i in ids:
        o = Instance3D(i, kf_id=i, points_ids=[], mask_area=100 + i, bank=ovo.bank)
        o.update([], i + 8, 50 + i)
        ovo.objects[i] = o
    ovo.update_objects_clip(force_update=True)
    before = {i: ovo.objects[i].clip_feature.cpu().numpy().reshape(-1).copy() for i in ids}
    kept, fused, ref_ins = OS.merge_instances(xyz, ins, ids, before, ovo.th_centroid, ovo.th_cossim, ovo.th_points)
    assert fused == {1: 0, 3: 2} and kept == [0, 2, 4, 5, 6], "the fixture must exercise both merge clauses and all rejections"
    d_ins = torch.from_numpy(ins).to(DEV)
    out = ovo.update_map((torch.from_numpy(xyz).to(DEV), None, d_ins), kfs=list(range(0, 160, 10)))
    assert np.array_equal(out.cpu().numpy(), ref_ins) and out.data_ptr() == d_ins.data_ptr()      # relabelled in place
    assert list(ovo.objects) == kept
    o0 = ovo.objects[0]
    assert o0.kfs_ids == [0, 8, 1, 9] and sorted(k for _, k in o0.top_kf) == [0, 1, 8, 9]
    # (with k_top_views = 0 the merged heap stays empty, add_top_kf never fires and -- like the reference -- the descriptor is
    #  not refreshed: instance3d.py:105-134)
    for kf, moved in ((1, 0), (9, 0), (3, 2), (11, 2)):                       # descriptors re-keyed to the surviving id
        view = ovo.keyframes["ins_descriptors"][kf]
        assert moved in view and len(view) == 1
    # avg_pooling over the union of views: (f0 + 0.5 f0 + f1 + 0.5 f1) / 4
    want = (1.5 * feats[0] + 1.5 * feats[1]) / 4
    np.testing.assert_allclose(o0.clip_feature.cpu().numpy().reshape(-1), want, atol=1e-6)
    np.testing.assert_allclose(ovo.objects[4].clip_feature.cpu().numpy().reshape(-1), before[4], atol=0)


def test_update_map_deleted_keyframes_and_noop():
    from ovo_amd.entities.descriptor_bank import KeyframeView
    from ovo_amd.entities.instance3d import Instance3D
    from ovo_amd.entities.ovo import OVO
    cfg = {"clip": {"k_top_views": 0, "fusion": "avg_pooling"}, "sam": {"precomputed": True}}
    ovo = OVO(cfg, None, "scene", torch.eye(3).to(DEV), device=DEV, clip_generator=_NoClip(), mask_generator=object())
    pts = torch.rand(100, 3, device=DEV)
    ins = torch.full((100,), -1, dtype=torch.int32, device=DEV)
    assert ovo.update_map((pts, None, ins), kfs=[]) is ins                     # no objects: nothing to do
    ovo.keyframes["frame_id"] = [0, 10, 20]
    row = ovo.bank.append(torch.ones(1, 32, device=DEV))[0]
    ovo.keyframes["ins_descriptors"][10] = KeyframeView(ovo.bank, {0: row})    # keyed 10: dropped with frame 10 (ovo.py:373-378)
    ovo.objects[0] = Instance3D(0, kf_id=10, points_ids=[], mask_area=5, bank=ovo.bank)
    ovo.objects[0].clip_feature = torch.ones(1, 32, device=DEV)
    ins[:50] = 0
    out = ovo.update_map((pts, None, ins), kfs=[0, 20])
    assert ovo.keyframes["frame_id"] == [0, "Deleted", 20] and 10 not in ovo.keyframes["ins_descriptors"]
    assert list(ovo.objects) == [0] and torch.equal(out, ins)


@pytest.mark.parametrize("tag", ["kd", "table"])
def test_update_map_vs_reference_golden(tag):
    """OVO.update_map against what the reference's own update_map / fuse_instances produced on the same scene
    (tests/golden/loopclose.npz, tools/gen_golden.py:gen_loopclose): relabelled map, surviving instances in order, their keyframe
    lists, top-k heaps and re-fused descriptors, the re-keyed per-keyframe descriptor tables, the frame-id list.  `kd`: the geometric
    predicate runs on the GPU (ovo_near_fraction vs the reference's same_instance over an exact nearest-neighbour stand-in for Open3D);
    `table`: the predicate is a lookup table on both sides (control flow only), with two keyframes deleted."""
    from conftest import golden
    from ovo_amd.entities.descriptor_bank import KeyframeView
    from ovo_amd.entities.instance3d import Instance3D
    from ovo_amd.entities.ovo import OVO
    d = golden("loopclose")
    feats = d["feats"]
    cfg = {"match_distance_th": 0.05, "track_th": 40, "clip": {"k_top_views": int(d["n_top_kf"]), "fusion": "avg_pooling"}, "sam": {"precomputed": True}}
    ovo = OVO(cfg, None, "scene", torch.eye(3).to(DEV), device=DEV, clip_generator=_NoClip(), mask_generator=object())
    ovo.th_centroid, ovo.th_cossim, ovo.th_points = (float(v) for v in d["th"])
    ovo.keyframes["frame_id"] = d["frame_ids"].tolist()
    for kf in range(16):
        i = kf % 8
        row = ovo.bank.append(torch.from_numpy(feats[i] * (1.0 if kf < 8 else 0.5))[None].to(DEV))[0]
        ovo.keyframes["ins_descriptors"][kf] = KeyframeView(ovo.bank, {i: row})
    for i in range(8):
        o = Instance3D(i, kf_id=i, points_ids=[], mask_area=100 + i, bank=ovo.bank)
        o.update([], i + 8, 50 + i)
        ovo.objects[i] = o
    ovo.update_objects_clip(force_update=True)
    before = np.stack([ovo.objects[i].clip_feature.cpu().numpy().reshape(-1) for i in range(8)])
    np.testing.assert_allclose(before, d[f"{tag}_before"], atol=1e-6)
    pairs = {tuple(p) for p in d["table_pairs"].tolist()}
    same = (lambda a, b: (a, b) in pairs) if tag == "table" else None
    ins = torch.from_numpy(d["ins"].copy()).to(DEV)
    out = ovo.update_map((torch.from_numpy(d["xyz"]).to(DEV), None, ins), kfs=d[f"{tag}_kfs"].tolist(), same_instance=same)
    assert np.array_equal(out.cpu().numpy(), d[f"{tag}_out_ins"])
    kept = d[f"{tag}_kept"].tolist()
    assert list(ovo.objects) == kept
    assert [str(f) for f in ovo.keyframes["frame_id"]] == d[f"{tag}_frame_id"].tolist()
    keys = sorted((kf, i) for kf, view in ovo.keyframes["ins_descriptors"].items() for i in view)
    assert keys == [tuple(r) for r in d[f"{tag}_desc_keys"].tolist()]
    for i in kept:
        o = ovo.objects[i]
        assert o.kfs_ids == d[f"{tag}_obj{i}_kfs"].tolist()
        assert sorted(o.top_kf) == [tuple(r) for r in d[f"{tag}_obj{i}_topkf"].tolist()]
        np.testing.assert_allclose(o.clip_feature.cpu().numpy().reshape(-1), d[f"{tag}_obj{i}_clip"], atol=1e-6)
