"""The CPU oracle against the reference's own outputs (tests/golden/*.npz from tools/gen_golden.py).

Integer / index results must be bit-exact; floating-point results within the tolerance written here.
"""
import numpy as np
import pytest

from conftest import golden, unpack
from oracle import features as OF
from oracle import geometry as OG
from oracle import semantic as OS


@pytest.mark.parametrize("t", [1, 2])
def test_geometry_bit_exact(t):
    d = golden(f"geometry_t{t}")
    assert np.array_equal(OG.frustum_corners(d["depth"], d["c2w"], d["K"]), d["corners"])
    ids = OG.frustum_point_ids(d["pts"], d["corners"])
    assert np.array_equal(ids, d["frustum_ids"])
    fp = d["pts"][ids]
    hom = np.hstack([fp, np.ones((len(fp), 1), np.float32)])
    assert np.array_equal(OG.project(hom, d["K"], d["w2c"]), d["project_uv"])
    mi, muv = OG.match(d["depth"], d["w2c"], fp, d["K"], float(d["th"]))
    assert np.array_equal(mi, d["match_idx"]) and np.array_equal(muv, d["match_uv"])


def test_vanilla_mapper_bit_exact():
    d = golden("vanilla_mapper")
    pm = OS.PointMap(d["K"])
    for i in range(3):
        pm.integrate(d[f"rgb{i}"], d[f"depth{i}"], d[f"c2w{i}"])
        assert pm.xyz.shape[0] == int(d[f"n{i}"])
    assert np.array_equal(pm.xyz, d["pcd"])            # fp32 coordinates, bit-exact (FMA chain)
    assert np.array_equal(pm.ids, d["pcd_ids"])
    assert np.array_equal(pm.rgb, d["pcd_colors"])
    assert (d["pcd_obj_ids"] == -1).all()


@pytest.mark.parametrize("tag,filt", [("nofilter", False), ("filter", True), ("ratio", True)])
def test_tracking_bit_exact(tag, filt):
    d = golden(f"tracking_{tag}")
    w = int(d["mask_w"])
    ratio = tuple(d["ratio"].tolist())
    ratio = (ratio[0], ratio[1], int(ratio[2])) if ratio else ()
    pm = OS.PointMap(d["K"])
    tr = OS.SemanticTracker(d["K"], 0.05, int(d["track_th"]), filt, int(d["n_top_views"]))
    for i in range(4):
        pm.integrate(d[f"rgb{i}"], d[f"depth{i}"], d[f"c2w{i}"])
        assert pm.xyz.shape[0] == int(d[f"pcd_n{i}"])
        assert np.array_equal(pm.ins, d[f"ins_before{i}"])
        masks = unpack(d[f"masks{i}"], w)
        matched, fused, n_matched, updated = tr.step(d[f"depth{i}"], ratio, pm.xyz, pm.ids, pm.ins,
                                                     d[f"c2w{i}"], d[f"seg{i}"], masks)
        pm.ins = updated
        assert n_matched == int(d[f"n_matched{i}"])
        assert matched == d[f"matched_ins_ids{i}"].tolist()
        assert np.array_equal(updated, d[f"updated{i}"])
        assert np.array_equal(fused, unpack(d[f"bmaps{i}"], w))
        assert tr.next_ins == int(d[f"next_ins_id{i}"])
    assert sorted(tr.objects) == d["obj_ids"].tolist()
    for j in tr.objects:
        o = tr.objects[j]
        assert o.kfs == d[f"obj{j}_kfs"].tolist()
        assert o.points == d[f"obj{j}_points"].reshape(-1).tolist()
        assert sorted(o.heap) == [tuple(r) for r in d[f"obj{j}_topkf"].tolist()]


def test_fusion():
    d = golden("fusion")
    rows = d["clips"][0]
    a, ka = OS.fuse_views(rows, "l1_medoid")
    b, kb = OS.fuse_views(rows, "cossim_medoid")
    c, _ = OS.fuse_views(rows, "avg_pooling")
    assert np.array_equal(a, d["l1"]) and np.array_equal(b, d["cos"]) and kb == int(d["cos_kf"])
    assert np.array_equal(c, d["avg"])
    rec = OS.InstanceRecord(5, 3)
    feats = {}
    for kf, area in enumerate(d["areas"].tolist()):
        rec.observe([kf * 10], kf, area)
        feats[kf] = {5: d["feats"][kf]}
        rec.refresh_feature(feats, "avg_pooling")
        np.testing.assert_allclose(rec.feature.reshape(-1), d["trace"][kf], rtol=0, atol=1e-7)
    assert sorted(rec.heap) == [tuple(r) for r in d["top_kf"].tolist()]


def test_similarity_and_crop_fusion():
    d = golden("similarity")
    np.testing.assert_allclose(OF.similarity(d["F"], d["T"]), d["clip"], atol=2e-6, rtol=0)
    s = OF.similarity(d["F"], d["T"], True, float(d["logit_scale"][0]), float(d["logit_bias"]))
    np.testing.assert_allclose(s, d["siglip"], atol=2e-6, rtol=0)
    for mode in ("fixed_weights", "hovsg", "adaptive_weights", "concept_fusion", "vanilla"):
        out = OF.fuse_crop_descriptors(d["cg"], d["cs"], d["cb"], mode, 0.4418, 0.1)
        np.testing.assert_allclose(out, d["fuse_" + mode], atol=1e-6, rtol=0)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_textregion_pooling(tag):
    d = golden("textregion")
    gh, gw, nh, nw = d[f"{tag}_grid"].tolist()
    P = int(d["crop"]) // int(d["patch"])
    masks = unpack(d[f"{tag}_masks"], int(d[f"{tag}_mask_w"]))
    fm = OF.feature_masks(masks, gh, gw)
    np.testing.assert_allclose(fm, d[f"{tag}_feature_masks"], atol=1e-6, rtol=0)
    x = OF.stitch_tokens(d[f"{tag}_tokens"][:, 1:], P, gh, gw, nh, nw)
    np.testing.assert_allclose(x, d[f"{tag}_x_input"][0], atol=1e-6, rtol=0)
    D = x.shape[1]
    w, b = d["in_proj_weight"], d["in_proj_bias"]
    out = OF.region_pool(x, fm, w[2 * D:], b[2 * D:], d["out_proj_weight"], d["out_proj_bias"], d["proj"])
    np.testing.assert_allclose(out, d[f"{tag}_out"], atol=2e-6, rtol=0)   # masked-mean identity


@pytest.mark.parametrize("tag", ["c", "d"])
def test_textregion_remove_global_patch(tag):
    """textregion.py:31-50 run by the reference itself (remove_global_patch=True, threshold 0.07): the filtered feature masks and the
    descriptors pooled from them pin the oracle's restatement (row a18)."""
    d = golden("textregion")
    gh, gw, nh, nw = d[f"{tag}_grid"].tolist()
    P = int(d["crop"]) // int(d["patch"])
    masks = unpack(d[f"{tag}_masks"], int(d[f"{tag}_mask_w"]))
    fm = OF.feature_masks(masks, gh, gw)
    np.testing.assert_allclose(fm, d[f"{tag}_feature_masks"], atol=1e-6, rtol=0)
    x = OF.stitch_tokens(d[f"{tag}_tokens"][:, 1:], P, gh, gw, nh, nw)
    np.testing.assert_allclose(x, d[f"{tag}_x_input"][0], atol=1e-6, rtol=0)
    kept, diff = OF.remove_global_patch(x, fm, float(d[f"{tag}_th"]))
    assert np.array_equal(kept, d[f"{tag}_kept_masks"])
    assert ((kept > 0).any(0) != (fm > 0).any(0)).sum() > 0                    # the filter really removed columns
    D = x.shape[1]
    w, b = d["in_proj_weight"], d["in_proj_bias"]
    out = OF.region_pool(x, kept, w[2 * D:], b[2 * D:], d["out_proj_weight"], d["out_proj_bias"], d["proj"])
    np.testing.assert_allclose(out, d[f"{tag}_out"], atol=2e-6, rtol=0)


@pytest.mark.parametrize("tag", ["kd", "table"])
def test_loop_closure_merge(tag):
    """ovo.py:381-407 + instance_utils.py:5-35 run by the reference (tools/gen_golden.py: gen_loopclose): surviving instances in
    order, the merge map and the relabelled per-point ids.  `kd`: the reference's own same_instance with an exact nearest-neighbour
    stand-in for Open3D; `table`: same_instance replaced by a lookup table (control flow only)."""
    d = golden("loopclose")
    ids = list(range(8))
    feats = {i: d[f"{tag}_before"][i] for i in ids}
    th = d["th"].tolist()
    pairs = {tuple(p) for p in d["table_pairs"].tolist()}
    same = (lambda a, b: (a, b) in pairs) if tag == "table" else None
    kept, fused, out = OS.merge_instances(d["xyz"], d["ins"], ids, feats, th[0], th[1], th[2], same=same)
    assert kept == d[f"{tag}_kept"].tolist()
    assert np.array_equal(out, d[f"{tag}_out_ins"])
    assert sorted(set(ids) - set(kept) - {7}) == sorted(fused)                  # 7 lost its points, the others were merged


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_mask_nms_and_segmap(tag):
    d = golden(f"segment_{tag}")
    masks = unpack(d["masks"], int(d["mask_w"]))
    keep = OF.mask_nms(masks, d["stability"] * d["pred_iou"])
    assert keep.tolist() == d["keep"].tolist()
    kept = sorted(keep.tolist())                                  # filter(): original order
    seg, bm = OF.paint_segmap(masks[kept], d["stability"][kept])
    assert np.array_equal(seg, d["seg_map"])
    assert np.array_equal(bm, unpack(d["bmaps"], int(d["mask_w"])))
    assert np.array_equal(OF.masks_to_boxes(masks), d["boxes"])


def test_query_and_classify():
    d = golden("query")
    rows = d["table"][d["tok_ids"]]                              # [Q, templates, D]
    T2 = OF.text_embeddings(rows)
    T1 = OF.text_embeddings(rows[:, :1])
    np.testing.assert_allclose(OF.similarity(d["feats"], T1), d["sim_single"], atol=2e-6, rtol=0)
    sim = OF.similarity(d["feats"], T2)
    np.testing.assert_allclose(sim, d["sim_ensemble"], atol=2e-6, rtol=0)
    cls, conf = OF.classify(d["sim_ensemble"], float(d["th"]))
    assert np.array_equal(cls, d["classes"])
    np.testing.assert_allclose(conf, d["conf"], atol=0, rtol=0)


def test_sam2_decoder_vs_hf_golden():
    """oracle/sam2_decoder.py (prompt encoder + two-way transformer + hyper-network heads) against HuggingFace's
    Sam2PromptEncoder / Sam2MaskDecoder on the same random weights (tools/gen_hf_sam2_decoder.py)."""
    import torch
    from oracle import sam2_decoder as SD
    d = golden("hf_sam2_decoder")
    sd = SD.hf_sam2_decoder_to_sam2({k[2:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("w:")})
    size = int(d["image_size"])
    sparse = SD.embed_points(sd, torch.from_numpy(d["points"]), torch.from_numpy(d["labels"]), size)
    np.testing.assert_allclose(sparse.numpy(), d["sparse"], atol=2e-6, rtol=0)
    s = d["embed"].shape[-1]
    ipe = SD.image_pe(sd, s).t().reshape(-1, s, s)
    np.testing.assert_allclose(ipe.numpy(), d["image_pe"], atol=2e-6, rtol=0)
    masks, iou, obj = SD.mask_decoder(sd, torch.from_numpy(d["embed"]), torch.from_numpy(d["feat_s1"]), torch.from_numpy(d["feat_s0"]),
                                      sparse, heads=int(d["heads"]), multimask=True)
    scale = np.abs(d["masks_multi"]).max()
    assert np.abs(masks.numpy() - d["masks_multi"]).max() < 2e-5 * max(scale, 1.0)
    np.testing.assert_allclose(iou.numpy(), d["iou_multi"], atol=2e-6, rtol=0)
    np.testing.assert_allclose(obj.numpy(), d["obj_multi"], atol=1e-5, rtol=0)


@pytest.mark.parametrize("act", ["quick_gelu", "gelu"])
def test_text_tower_vs_hf_golden(act):
    """oracle/text.py (CLIP text transformer, causal mask, end-of-text pooling) against HuggingFace's
    CLIPTextModelWithProjection on the same random weights (tools/gen_hf_text.py)."""
    import torch
    from oracle import text as OT
    d = golden("hf_clip_text")
    pre = f"{act}:w:"
    sd = OT.hf_clip_text_to_openclip({k[len(pre):]: torch.from_numpy(d[k]) for k in d.files if k.startswith(pre)})
    out = OT.text_forward(sd, torch.from_numpy(d[f"{act}:ids"]), heads=int(d["heads"]), act=act)
    np.testing.assert_allclose(out.numpy(), d[f"{act}:out"], atol=3e-6, rtol=1e-5)


def test_siglip_tower_vs_hf_golden():
    """oracle/vit.py on the SigLIP variant (no class token, no pre-LN, tanh-GELU, attention-pooling head) against
    HuggingFace's SiglipVisionModel on the same random weights (tools/gen_hf_siglip.py)."""
    import torch
    from oracle import vit as OV
    d = golden("hf_siglip_vit")
    sd = OV.hf_siglip_to_openclip({k[2:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("w:")})
    x = torch.from_numpy(d["x"])
    tok = OV.vit_forward(sd, x, patch=int(d["patch"]), heads=int(d["heads"]), act="gelu_tanh", pre_ln=False, cls_token=False, eps=1e-6, tokens=True)
    np.testing.assert_allclose(tok.numpy(), d["tokens"], atol=2e-5, rtol=1e-5)
    pooled = OV.map_pool(sd, tok, heads=int(d["heads"]))
    np.testing.assert_allclose(pooled.numpy(), d["pooled"], atol=2e-5, rtol=1e-5)


def test_siglip_text_tower_vs_hf_golden():
    """oracle/text.py in SigLIP mode (no causal mask, last-position pooling, tanh-GELU, projection with bias) against
    HuggingFace's SiglipTextModel on the same random weights (tools/gen_hf_siglip_text.py)."""
    import torch
    from oracle import text as OT
    d = golden("hf_siglip_text")
    sd = OT.hf_siglip_text_to_openclip({k[2:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("w:")})
    out = OT.text_forward(sd, torch.from_numpy(d["ids"]), heads=int(d["heads"]), act="gelu_tanh", eps=1e-6, causal=False, pool="last")
    np.testing.assert_allclose(out.numpy(), d["out"], atol=3e-6, rtol=1e-5)
