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
