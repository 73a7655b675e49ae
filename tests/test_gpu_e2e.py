"""-m gpu: the reference's usage flow (OVOSemMap.run, ovomapping.py:139-187, then run_eval's query / checkpoint) with every
stage native: SAM2 encoder + mask decoder + automatic mask generator -> tracking on the point map -> TextRegion descriptors ->
multi-view fusion -> text tower -> query -> checkpoint round trip.  Small random-weight models; what is checked is that the
stages compose through the reference's API and that the state survives the reference's on-disk format."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _tokenizer(context):
    def tok(texts):
        rows = []
        for s in texts:
            ids = [98] + [1 + (sum(map(ord, w)) % 90) for w in s.split()][: context - 2] + [99]
            rows.append(ids + [0] * (context - len(ids)))
        return torch.tensor(rows)
    return tok


def test_full_native_flow_and_checkpoint(tmp_path):
    from ovo_amd import synthetic as syn
    from ovo_amd.encoders.text import HipTextEncoder, TextSpec
    from ovo_amd.encoders.vit import SPECS as VS, HipViT
    from ovo_amd.entities.clip_generator import CLIPGenerator
    from ovo_amd.entities.mask_generator import MaskGenerator
    from ovo_amd.entities.ovo import OVO
    from ovo_amd.slam.vanilla_mapper import VanillaMapper
    from ovo_amd.utils import io_utils
    scale = 0.35
    K = torch.from_numpy(syn.scannet_intrinsics(scale)).to(DEV)
    vit = HipViT(VS["tiny-pe"], None, device=DEV, seed=1)
    text = HipTextEncoder(TextSpec("tiny", 100, 16, 64, 2, 4, VS["tiny-pe"].out_dim, "gelu"), None, device=DEV, tokenizer=_tokenizer(16))
    clip_cfg = {"embed_type": "TextRegion", "model_card": "PE-tiny-084", "k_top_views": 5, "fusion": "l1_medoid"}
    sam_cfg = {"sam_encoder": "hiera_test256", "sam_decoder": "sam2_small", "points_per_side": 6, "nms_iou_th": 0.45, "stability_score_th": 0.5,
               "nms_score_th": 0.2, "nms_inner_th": 0.5, "seed": 1}
    cfg = {"match_distance_th": 0.05, "track_th": 30, "depth_filter": True, "log": False, "kf_queue_delay": 1, "debug_info": True,
           "clip": clip_cfg, "sam": sam_cfg}

    def build():
        gen = CLIPGenerator(dict(clip_cfg), device=DEV, encoder=vit, text_encoder=text)
        mg = MaskGenerator(dict(sam_cfg), None, device=DEV)
        mg.mask_generator.box_nms_thresh = 1.0            # random weights: identical boxes would all be suppressed
        return OVO(dict(cfg), None, None, K, device=DEV, clip_generator=gen, mask_generator=mg)
    ovo = build()
    vm = VanillaMapper({"device": DEV, "mapping": {}}, K)
    n_masks = []
    for t in range(5):                                     # OVOSemMap.run's loop body, every frame a keyframe
        fid, rgb, depth, c2w = syn.frame(t, scale=scale, seed=3)
        fd = [fid, rgb, depth, c2w]
        vm.track_camera(fd)
        pose = vm.get_c2w(fid)
        vm.map(fd, pose)
        updated = ovo.detect_and_track_objects([fid, rgb, depth, ()], vm.get_map(), pose)
        if updated is not None:
            vm.update_pcd_obj_ids(updated)
            n_masks.append(int(ovo.keyframes_queue[-1][1].shape[0]) if ovo.keyframes_queue else 0)
        ovo.compute_semantic_info()                        # kf_queue_delay = 1: one keyframe behind
    ovo.complete_semantic_info()
    assert len(ovo.objects) > 0 and len(ovo.keyframes_queue) == 0, f"no instance tracked (masks per frame {n_masks})"
    ids = vm.get_map()[2]
    assert ids.dtype == torch.int32 and int((ids >= 0).sum()) > 0 and set(torch.unique(ids[ids >= 0]).tolist()) <= set(ovo.objects)

    classes = ["chair", "a wooden table", "lamp", "floor"]
    sim = ovo.query(classes)
    info = ovo.classify_instances(classes, th=-1.0)
    n_obj = len(ovo.objects)
    with_desc = [i for i, o in enumerate(ovo.objects.values()) if o.clip_feature is not None]
    assert sim.shape == (n_obj, len(classes)) and info["classes"].shape == (n_obj,) and len(with_desc) > 0
    s = sim.cpu().numpy()[with_desc]
    assert np.isfinite(s).all() and np.array_equal(info["classes"][with_desc], s.argmax(1))

    # checkpoint: the reference's ovo_map.ckpt layout, through its writer, and back into fresh objects
    ckpt = {"map_params": vm.get_map_dict(), "ovo_map_params": ovo.capture_dict(debug_info=True)}
    io_utils.save_dict_to_ckpt(ckpt, "ovo_map.ckpt", directory=tmp_path)
    back = torch.load(tmp_path / "ovo_map.ckpt", map_location="cpu", weights_only=False)
    vm2 = VanillaMapper({"device": DEV, "mapping": {}}, K)
    vm2.set_map_dict(back["map_params"])
    assert torch.equal(vm2.get_map()[0], vm.get_map()[0]) and torch.equal(vm2.get_map()[2], vm.get_map()[2])
    ovo2 = build()
    ovo2.restore_dict(back["ovo_map_params"], debug_info=True)
    assert list(ovo2.objects) == list(ovo.objects)
    sim2 = ovo2.query(classes)
    assert torch.equal(torch.nan_to_num(sim2), torch.nan_to_num(sim))
    # per-vertex predictions in the reference's files
    labels = np.full(ids.shape[0], -1, np.int64)
    obj_ids = list(ovo.objects)
    ids_h = ids.cpu().numpy()
    for k, oid in enumerate(obj_ids):
        labels[ids_h == oid] = info["classes"][k]
    io_utils.write_labels(str(tmp_path / "labels.txt"), labels)
    assert np.array_equal(io_utils.read_labels(str(tmp_path / "labels.txt")), labels)
    masks = np.stack([ids_h == oid for oid in obj_ids]).astype(np.uint8)
    io_utils.write_instances(str(tmp_path), "scene", {"masks": masks, "classes": info["classes"], "conf": info["conf"]})
    lines = (tmp_path / "instance_pred" / "scene.txt").read_text().splitlines()
    assert len(lines) == n_obj and all(len(l.split()) == 3 for l in lines)


class _SyntheticDataset:
    """The reference's dataset protocol (datasets.py:79): indexable frame tuples + camera attributes."""

    def __init__(self, n, scale, seed):
        from ovo_amd import synthetic as syn
        self.n, self.scale, self.seed, self.syn = n, scale, seed, syn
        self.intrinsics = syn.scannet_intrinsics(scale)
        self.height, self.width = syn.scannet_depth_hw(scale)
        self.blank = set()

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        fid, rgb, depth, c2w = self.syn.frame(i, scale=self.scale, seed=self.seed)
        if i in self.blank:
            depth = np.zeros_like(depth)
        return fid, rgb, depth, c2w


def test_ovosemmap_driver_run_checkpoint_and_resume(tmp_path):
    """OVOSemMap (ovomapping.py:29-243): cadence (map every frame, segment every 2nd), skip of a frame without depth, logger files,
    ovo_map.ckpt + estimated_c2w.npy, and `restore_map` resuming after the last stored pose -- against the loop written out by hand."""
    from ovo_amd.encoders.vit import SPECS as VS, HipViT
    from ovo_amd.entities.clip_generator import CLIPGenerator
    from ovo_amd.entities.mask_generator import MaskGenerator
    from ovo_amd.entities.ovo import OVO
    from ovo_amd.entities.ovomapping import OVOSemMap
    from ovo_amd.slam.vanilla_mapper import VanillaMapper
    scale, n = 0.35, 7
    ds = _SyntheticDataset(n, scale, 3)
    ds.blank.add(3)
    K = torch.from_numpy(ds.intrinsics).to(DEV)
    vit = HipViT(VS["tiny-pe"], None, device=DEV, seed=1)
    clip_cfg = {"embed_type": "TextRegion", "model_card": "PE-tiny-084", "k_top_views": 5, "fusion": "l1_medoid"}
    sam_cfg = {"sam_encoder": "hiera_test256", "sam_decoder": "sam2_small", "points_per_side": 6, "nms_iou_th": 0.45, "stability_score_th": 0.5,
               "nms_score_th": 0.2, "nms_inner_th": 0.5, "seed": 1}
    sem = {"segment_every": 2, "match_distance_th": 0.05, "track_th": 30, "depth_filter": True, "log": True, "kf_queue_delay": 1,
           "clip": clip_cfg, "sam": sam_cfg}

    def build(logger=None):
        import copy
        mg = MaskGenerator(dict(sam_cfg), None, device=DEV)
        mg.mask_generator.box_nms_thresh = 1.0
        return OVO(copy.deepcopy(sem), logger, None, K, device=DEV, clip_generator=CLIPGenerator(dict(clip_cfg), device=DEV, encoder=vit), mask_generator=mg)

    def config(**extra):
        import copy
        return {"device": DEV, "dataset_name": "synthetic", "vis": {"stream": False, "show_stream": False}, "mapping": {"map_every": 1},
                "semantic": copy.deepcopy(sem), "slam": {"slam_module": "vanilla", "save_estimated_cam": True}, "use_wandb": False,
                "data": {"scene_name": "scene0000_00"}, **extra}
    run = OVOSemMap(config(), str(tmp_path / "run"), dataset=ds, ovo=build())
    run.run()

    # the same sequence by hand
    ovo, vm = build(), VanillaMapper({"device": DEV, "mapping": {}}, K)
    for t in range(n):
        fd = list(ds[t])
        vm.track_camera(fd)
        if t in ds.blank:
            continue
        pose = vm.get_c2w(t)
        vm.map(fd, pose)
        if t % 2 == 0:
            up = ovo.detect_and_track_objects([t, fd[1], fd[2], ()], vm.get_map(), pose)
            if up is not None:
                vm.update_pcd_obj_ids(up)
            ovo.compute_semantic_info()
    ovo.complete_semantic_info()
    a, b = run.slam_backbone.get_map(), vm.get_map()
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]) and list(run.ovo.objects) == list(ovo.objects) and len(ovo.objects) > 0
    for k in ovo.objects:
        fa, fb = run.ovo.objects[k].clip_feature, ovo.objects[k].clip_feature
        assert (fa is None) == (fb is None) and (fa is None or torch.equal(fa, fb))

    out = tmp_path / "run"
    assert (out / "config.yaml").exists() and (out / "ovo_map.ckpt").exists() and (out / "estimated_c2w.npy").exists()
    logs = {p.name: p.read_text().splitlines() for p in (out / "logger").glob("*.log")}
    assert sorted(logs["frame_id.log"]) == ["0", "0", "2", "2", "4", "4", "6", "6"] and len(logs["t_sam.log"]) == 4     # tracking + (one keyframe later) descriptors
    assert len(logs["vram.log"]) == 4 and len(logs["avg_fps.log"]) == 1 and "n_obj.log" not in logs and float(logs["max_vram.log"][0]) > 0

    # resume: a longer sequence continues after the last stored pose and ends where an uninterrupted run ends
    ds2 = _SyntheticDataset(n + 4, scale, 3)
    ds2.blank.add(3)
    resumed = OVOSemMap(config(restore_map=True), str(out), dataset=ds2, ovo=build())
    assert resumed.first_frame == n and list(resumed.ovo.objects) == list(ovo.objects)
    assert torch.equal(resumed.slam_backbone.get_map()[0], b[0])
    resumed.run()
    full = OVOSemMap(config(), str(tmp_path / "full"), dataset=ds2, ovo=build())
    full.run()
    assert torch.equal(resumed.slam_backbone.get_map()[0], full.slam_backbone.get_map()[0])
    assert resumed.slam_backbone.get_map()[0].shape[0] > b[0].shape[0]
    with pytest.raises(NotImplementedError):
        OVOSemMap(config(slam={"slam_module": "orbslam2"}), str(tmp_path / "x"), dataset=ds, ovo=build())
    with pytest.raises(NotImplementedError):
        OVOSemMap(config(vis={"stream": True}), str(tmp_path / "y"), dataset=ds, ovo=build())
