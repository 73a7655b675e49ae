"""-m gpu: descriptor fusion, dense scatter-accumulate, similarity query and mask NMS kernels vs the oracle.

Floating-point tolerances are written next to each comparison; integer results are bit-exact.
"""
import numpy as np
import pytest
import torch

from conftest import golden, unpack

pytestmark = pytest.mark.gpu


def _bound(out_dim: int) -> float:
    """north_star's feature bound: |unit-descriptor error| <= 1e-3, flat, for every model the reference can select (out_dim >= 512);
    the reduced test towers (out_dim 64 / 128) have proportionally larger elements, 1/sqrt(out_dim)."""
    return 1e-3 if out_dim >= 512 else 1e-3 * (512 / out_dim) ** 0.5 * 1.5
DEV = "cuda"


def _t(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t if dtype is None else t.to(dtype)).to(DEV)


def test_similarity_golden_and_argmax():
    from ovo_amd.utils import clip_utils as CU
    d = golden("similarity")
    F, T = _t(d["F"]), _t(d["T"])
    np.testing.assert_allclose(CU.clip_cosine_similarity(T, F).cpu().numpy(), d["clip"], atol=1e-6, rtol=0)
    s = CU.siglip_cosine_similarity(T, F, float(d["logit_scale"][0]), float(d["logit_bias"]))
    np.testing.assert_allclose(s.cpu().numpy(), d["siglip"], atol=1e-6, rtol=1e-5)
    # fused argmax == argmax of the kernel's own scores (bit-exact given S), with threshold semantics of ovo.py:487-491
    sim, cls, conf = CU.similarity(F, T, want_argmax=True, th=0.25)
    sim = sim.cpu().numpy()
    ref_cls = sim.argmax(1)
    ref_conf = sim.max(1)
    ref_cls[ref_conf <= 0.25] = -1
    ref_conf[ref_conf <= 0.25] = 0
    assert np.array_equal(cls.cpu().numpy(), ref_cls) and np.array_equal(conf.cpu().numpy(), ref_conf)
    assert (ref_cls == -1).any() and (ref_cls >= 0).any()


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-6), (torch.float16, 1e-3), (torch.bfloat16, 8e-3)])
@pytest.mark.parametrize("q", [1, 10, 16, 17, 100])
def test_similarity_vs_oracle(dtype, tol, q):
    from oracle import features as OF
    from ovo_amd import synthetic as syn
    from ovo_amd.utils import clip_utils as CU
    n, d = 4099, 768
    F = syn.unit_vectors(n, d, seed=1)
    T = syn.unit_vectors(q, d, seed=2)
    Fd = _t(F).to(dtype)
    sim, cls, conf = CU.similarity(Fd, _t(T), want_argmax=True, th=-10.0)
    Tr = T if (dtype == torch.float32 or q < CU.LARGE_VOCABULARY) else torch.from_numpy(T).to(dtype).float().numpy()
    ref = OF.similarity(Fd.float().cpu().numpy(), Tr)        # oracle on the same (rounded) inputs: tolerance = accumulation only
    np.testing.assert_allclose(sim.cpu().numpy(), ref, atol=2e-6, rtol=0)
    np.testing.assert_allclose(sim.cpu().numpy(), OF.similarity(F, T), atol=tol, rtol=0)   # <= 1e-3 in fp16 (north_star)
    assert np.array_equal(cls.cpu().numpy(), sim.argmax(1).cpu().numpy())


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 1e-3), (torch.bfloat16, 8e-3)])
@pytest.mark.parametrize("q,siglip", [(64, False), (1000, False), (1001, False), (202, True)])
def test_large_vocabulary_query(dtype, tol, q, siglip):
    """BASELINE config 5's shape (1k texts x 16-bit map): MFMA GEMM + row argmax vs the f64-accumulating oracle."""
    from oracle import features as OF
    from ovo_amd import synthetic as syn
    from ovo_amd.utils import clip_utils as CU
    n, d = 20011, 768
    F = syn.unit_vectors(n, d, seed=3)
    T = syn.unit_vectors(q, d, seed=4)
    Fd = _t(F).to(dtype)
    ls, lb = (2.5, -1.0) if siglip else (0.0, 0.0)
    sim, cls, conf = CU.similarity(Fd, _t(T), want_argmax=True, th=-10.0, siglip=siglip, logit_scale=ls, logit_bias=lb)
    assert sim.shape == (n, q)
    Tr = torch.from_numpy(T).to(dtype).float().numpy()
    ref = OF.similarity(Fd.float().cpu().numpy(), Tr, siglip, ls, lb)
    np.testing.assert_allclose(sim.cpu().numpy(), ref, atol=3e-6, rtol=0)                    # same rounded inputs: accumulation only
    np.testing.assert_allclose(sim.cpu().numpy(), OF.similarity(F, T, siglip, ls, lb), atol=tol * (3 if siglip else 1), rtol=0)
    s = sim.cpu().numpy()
    assert np.array_equal(cls.cpu().numpy(), s.argmax(1)) and np.array_equal(conf.cpu().numpy(), s.max(1))
    # threshold semantics (ovo.py:487-491): rows whose best score is <= th get class -1, confidence 0
    th = float(np.median(s.max(1)))
    _, cls2, conf2 = CU.similarity(Fd, _t(T), want_argmax=True, th=th, siglip=siglip, logit_scale=ls, logit_bias=lb)
    below = s.max(1) <= th
    assert np.array_equal(cls2.cpu().numpy(), np.where(below, -1, s.argmax(1))) and np.all(conf2.cpu().numpy()[below] == 0)


def test_dense_query_row_scale():
    from oracle import features as OF
    from ovo_amd.utils import clip_utils as CU
    rng = np.random.default_rng(0)
    acc = rng.standard_normal((1000, 64)).astype(np.float32)
    cnt = rng.integers(0, 4, 1000).astype(np.int32)
    T = rng.standard_normal((10, 64)).astype(np.float32)
    sim = CU.similarity(_t(acc), _t(T), cnt=_t(cnt))[0].cpu().numpy()
    ref = OF.similarity(acc, T) * np.where(cnt > 0, 1.0 / np.maximum(cnt, 1), 0)[:, None].astype(np.float32)
    np.testing.assert_allclose(sim, ref, atol=1e-5, rtol=1e-5)


def test_fuse_views_golden_and_bank():
    from ovo_amd.entities.descriptor_bank import DescriptorBank
    from ovo_amd.entities.instance3d import Instance3D
    d = golden("fusion")
    rows = _t(d["clips"][0])
    bank = DescriptorBank(rows.shape[1], DEV, rows=2, slots=1)      # tiny capacities: exercise growth
    r = bank.append(rows)
    for mode, key, kfkey in (("avg_pooling", "avg", None), ("l1_medoid", "l1", None), ("cossim_medoid", "cos", "cos_kf")):
        kf = bank.fuse([(7, r), (9, r[:1])], mode)
        np.testing.assert_allclose(bank.feature(7).reshape(-1).cpu().numpy(), d[key], atol=1e-6, rtol=0)
        assert bank.feature(7).shape == (1, rows.shape[1]) and bank.feature(9).shape == (rows.shape[1],)
        assert kf[9] == 0
        if kfkey:
            assert kf[7] == int(d[kfkey])
    # Instance3D.update_clip flow with a top-3 heap (instance3d.py:105-189)
    Instance3D.n_top_kf = 3
    Instance3D.set_fusion("avg_pooling")
    from ovo_amd.entities.descriptor_bank import KeyframeView
    inst = Instance3D(5, bank=bank)
    views = {}
    for kf, area in enumerate(d["areas"].tolist()):
        inst.update([kf * 10], kf, area)
        views[kf] = KeyframeView(bank, {5: bank.append(_t(d["feats"][kf:kf + 1]))[0]})
        inst.update_clip(views)
        np.testing.assert_allclose(inst.clip_feature.reshape(-1).cpu().numpy(), d["trace"][kf], atol=1e-6, rtol=0)
    assert sorted(inst.top_kf) == [tuple(x) for x in d["top_kf"].tolist()]
    exp = inst.export(True)
    assert exp["ins3d_5_clip_feature"].device.type == "cpu" and exp["ins3d_5_keyframes_ids"].tolist() == [0, 1, 2, 3, 4]


def test_running_sum_fusion_vs_full_refusion():
    """`DescriptorBank.fuse_add` (avg_pooling as a running sum, `ovo_fuse_views_add`): after every batch of new views the table row equals the
    full re-fusion of all views in ARRIVAL order bit for bit (same additions in the same order), and the re-fusion in any other stacking
    order -- the heap's, largest area first -- to float rounding; restarting a sum (`before` = 0) forgets the old one."""
    from ovo_amd.entities.descriptor_bank import DescriptorBank
    g = torch.Generator().manual_seed(21)
    D = 96
    inc, full = DescriptorBank(D, DEV, rows=4, slots=2), DescriptorBank(D, DEV, rows=4, slots=2)
    seen = {3: [], 11: [], 40: []}
    for step in range(12):
        ups_inc, ups_full = [], []
        for ins in seen:
            n_new = int(torch.randint(0, 3, (1,), generator=g))
            if n_new == 0:
                continue
            feats = torch.randn(n_new, D, generator=g).to(DEV) * 3
            r_inc, r_full = inc.append(feats), full.append(feats)
            assert r_inc == r_full
            ups_inc.append((ins, r_inc, len(seen[ins])))
            seen[ins] = seen[ins] + r_full
            ups_full.append((ins, seen[ins]))
        inc.fuse_add(ups_inc)
        full.fuse(ups_full, "avg_pooling")
        for ins, rows in seen.items():
            if not rows:
                continue
            assert torch.equal(inc.feature(ins), full.feature(ins)) and inc.feature(ins).shape == full.feature(ins).shape
            other = DescriptorBank(D, DEV, rows=len(rows), slots=1)
            other.store, other.n_rows = full.store, full.n_rows
            other.fuse([(ins, rows[::-1])], "avg_pooling")
            torch.testing.assert_close(inc.feature(ins), other.feature(ins), atol=2e-6, rtol=0)
    rows = inc.append(torch.ones(1, D, device=DEV))
    inc.fuse_add([(3, rows, 0)])                                  # before = 0: the sum starts over
    assert torch.equal(inc.feature(3), torch.ones(D, device=DEV))


def test_scatter_accum_linearity():
    """acc += desc[row(seg)] for matched points only; two passes == 2x one pass; counts exact."""
    from ovo_amd import _lib as L
    rng = np.random.default_rng(4)
    n, D, n_masks = 50_001, 768, 32
    seg = rng.integers(-2, n_masks, n).astype(np.int16)
    mask_row = np.where(np.arange(n_masks) % 5 == 0, -1, rng.permutation(n_masks)).astype(np.int32)
    desc = rng.standard_normal((n_masks, D)).astype(np.float32)
    acc = torch.zeros((n, D), dtype=torch.float32, device=DEV)
    cnt = torch.zeros(n, dtype=torch.int32, device=DEV)
    lib = L.load()
    d_seg, d_row, d_desc = _t(seg), _t(mask_row), _t(desc)       # keep alive: only raw pointers cross the ABI
    for _ in range(2):
        L.check(lib.ovo_scatter_accum(L.ptr(d_seg), n, L.ptr(d_row), n_masks, L.ptr(d_desc), D, L.ptr(acc),
                                      L.ptr(cnt), L.stream()))
    rows = np.where(seg >= 0, mask_row[np.clip(seg, 0, None)], -1)
    hit = rows >= 0
    ref = np.zeros((n, D), np.float32)
    ref[hit] = 2 * desc[rows[hit]]
    assert np.array_equal(acc.cpu().numpy(), ref)              # x + x is exact
    assert np.array_equal(cnt.cpu().numpy(), 2 * hit.astype(np.int32))

@pytest.mark.parametrize("D,Q,shards", [(1024, 10, 1), (768, 16, 1), (400, 1, 1), (1024, 10, 4), (48, 3, 2)])
def test_scatter_query_fused_vs_two_passes(D, Q, shards):
    """ovo_scatter_accum_query (one launch from a hit list: accumulate + re-query) against ovo_scatter_accum_touched + ovo_similarity_rows through the
    C ABI: accumulators, counts, classes and confidences bit-identical; rows whose mask has no descriptor stay untouched; D % 32 == 16 tail."""
    from ovo_amd import _lib as L
    lib = L.load()
    rng = np.random.default_rng(D + Q)
    n, n_masks, blk = 70_003, 40, 256
    seg = rng.integers(-2, n_masks + 3, n).astype(np.int16)       # ids >= n_masks: masks the tracker does not know (skipped)
    mask_row = np.where(np.arange(n_masks) % 7 == 0, -1, rng.permutation(n_masks)).astype(np.int32)
    desc = rng.standard_normal((n_masks, D)).astype(np.float32)
    texts = rng.standard_normal((Q, D)).astype(np.float32)
    texts /= np.linalg.norm(texts, axis=1, keepdims=True)
    d_seg, d_row, d_desc, d_T = _t(seg), _t(mask_row), _t(desc), _t(texts)
    for rank in range(shards):
        nb = -(-n // blk)
        rows_local = -(-nb // shards) * blk if shards > 1 else n
        acc0 = torch.from_numpy(rng.standard_normal((rows_local, D)).astype(np.float32)).to(DEV)
        cnt0 = torch.from_numpy(rng.integers(0, 5, rows_local).astype(np.int32)).to(DEV)
        # the list the tracking pass would emit: every point with a segment id of this keyframe (also those whose mask has no descriptor), shuffled
        i = np.nonzero((seg >= 0) & (seg < n_masks))[0]
        if shards > 1:
            i = i[(i // blk) % shards == rank]
            local = (i // blk // shards) * blk + i % blk
        else:
            local = i
        hits = _t(rng.permutation(local).astype(np.int32))
        n_hits = torch.tensor([hits.numel()], dtype=torch.int32, device=DEV)
        out = []
        for fused in (True, False):
            acc, cnt = acc0.clone(), cnt0.clone()
            cls = torch.full((rows_local,), -7, dtype=torch.int64, device=DEV)
            conf = torch.full((rows_local,), -7.0, dtype=torch.float32, device=DEV)
            if fused:
                L.check(lib.ovo_scatter_accum_query(L.ptr(hits), L.ptr(n_hits), n, L.ptr(d_seg), L.ptr(d_row), n_masks, L.ptr(d_desc), D, L.ptr(acc), L.ptr(cnt),
                                                    rank, shards, blk, L.ptr(d_T), Q, 0, 0.0, 0.0, 0.05, L.ptr(cls), L.ptr(conf), L.stream()))
            else:
                touched = torch.empty(rows_local, dtype=torch.int32, device=DEV)
                n_t = torch.zeros(2, dtype=torch.int32, device=DEV)
                L.check(lib.ovo_scatter_accum_touched(L.ptr(d_seg), n, L.ptr(d_row), n_masks, L.ptr(d_desc), D, L.ptr(acc), L.ptr(cnt), L.ptr(touched),
                                                      n_t.data_ptr(), n_t[1:].data_ptr(), rank, shards, blk, L.stream()))
                L.check(lib.ovo_similarity_rows(L.ptr(acc), 0, L.ptr(touched), n_t.data_ptr(), rows_local, D, L.ptr(d_T), Q, L.ptr(cnt), 0, 0.0, 0.0, 0.05,
                                                L.ptr(cls), L.ptr(conf), L.stream())) if D % 16 == 0 else None
            out.append((acc, cnt, cls, conf))
        (a1, c1, k1, f1), (a2, c2, k2, f2) = out
        assert torch.equal(a1, a2) and torch.equal(c1, c2)
        assert torch.equal(k1, k2) and torch.equal(f1, f2)
        changed = (c1 != cnt0)
        assert int(changed.sum()) > 0 and int((k1 != -7).sum()) == int(changed.sum())
        live = mask_row[seg[i]] >= 0
        assert int(changed.sum()) == int(live.sum())

@pytest.mark.parametrize("q,dt", [(1000, torch.float16), (130, torch.bfloat16), (64, torch.float16)])
def test_large_vocabulary_scores_in_16_bits_with_fused_argmax(q, dt, monkeypatch):
    """BASELINE configs[4] with the scores it names (fp16): the 16-bit score matrix leaves the GEMM through the staged epilogue with the argmax still
    fused (round 6).  Classes / confidences equal the f32-score run's exactly (the maximum is taken before the rounding), the 16-bit scores are the
    rounded f32 scores, and the unstaged form (OVO_8P_BEST_STAGED=0) stores the same bits."""
    from ovo_amd.utils import clip_utils as CU
    monkeypatch.setenv("OVO_KNOBS_DYNAMIC", "1")
    g = torch.Generator(device=DEV).manual_seed(q)
    n, d = 70_001, 768
    F = torch.nn.functional.normalize(torch.randn((n, d), generator=g, device=DEV), dim=1).to(dt)
    T = torch.nn.functional.normalize(torch.randn((q, d), generator=g, device=DEV), dim=1)
    s32, c32, f32 = CU.similarity(F, T, want_argmax=True)
    s16, c16, f16 = CU.similarity(F, T, want_argmax=True, sim_dtype=dt)
    assert s16.dtype == dt and s16.shape == (n, q)
    assert torch.equal(c16, c32) and torch.equal(f16, f32)
    assert torch.equal(s16, s32.to(dt))
    monkeypatch.setenv("OVO_8P_BEST_STAGED", "0")
    s16b, c16b, f16b = CU.similarity(F, T, want_argmax=True, sim_dtype=dt)
    assert torch.equal(s16b, s16) and torch.equal(c16b, c16) and torch.equal(f16b, f16)
    ref = F.float() @ T.to(dt).float().t()
    assert float((s16.float() - ref).abs().max()) < 1e-3 * (1 if dt == torch.float16 else 4)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_mask_nms_golden(tag):
    from ovo_amd.utils import segment_utils as SU
    d = golden(f"segment_{tag}")
    masks = unpack(d["masks"], int(d["mask_w"]))
    keep = SU.mask_nms(_t(masks), _t(d["stability"] * d["pred_iou"]), iou_thr=0.8, score_thr=0.7, inner_thr=0.5)
    assert keep.cpu().tolist() == d["keep"].tolist()
    dicts = [{"segmentation": masks[i], "predicted_iou": d["pred_iou"][i], "stability_score": d["stability"][i]}
             for i in range(masks.shape[0])]
    kept, = SU.masks_update(dicts, iou_thr=0.8, score_thr=0.7, inner_thr=0.5)
    seg, bm = SU.mask2segmap(kept, np.zeros(masks.shape[1:] + (3,), np.uint8))
    assert seg.dtype == np.int32 and np.array_equal(seg, d["seg_map"])
    assert np.array_equal(bm, unpack(d["bmaps"], int(d["mask_w"])))
    assert np.array_equal(SU.batched_mask_to_box(_t(masks)).cpu().numpy(), d["boxes"])


def test_mask_intersections_full_res():
    from oracle import features as OF
    from ovo_amd import synthetic as syn
    from ovo_amd.utils import segment_utils as SU
    masks = syn.make_masks(480, 640, grid=(6, 8), n_blobs=40, seed=2)
    inter = SU.mask_intersections(_t(masks)).cpu().numpy()
    assert np.array_equal(inter, OF.mask_intersections(masks))


# ------------------------------------------------------------------ a14: per-mask crops + crop-mode descriptors
def _crop_fixture(H=150, W=200):
    from ovo_amd import synthetic as syn
    masks = syn.make_masks(H, W, grid=(2, 3), n_blobs=4, seed=2)
    extra = np.zeros((4, H, W), bool)
    extra[0, 40:90, 70] = True            # one pixel wide -> w = 0 after the inclusive-edge subtraction (degenerate)
    extra[1, 0:30, 0:45] = True           # touches the top-left corner (margin clamps at 0)
    extra[2, H - 20:H, W - 60:W] = True   # touches the bottom-right corner (margin slices at the image edge)
    # extra[3] stays empty -> box 0,0,0,0
    masks = np.concatenate([masks, extra])
    img = syn.render_rgb(H, W, 4).transpose(2, 0, 1).copy()
    return masks, img


@pytest.mark.parametrize("also_bbox", [False, True])
@pytest.mark.parametrize("as_u8", [True, False])
def test_mask_crops_vs_oracle(also_bbox, as_u8):
    from oracle import features as OF
    from ovo_amd.utils import segment_utils as SU
    masks, img = _crop_fixture()
    img = img if as_u8 else img.astype(np.float32) * 0.731
    boxes = SU.mask_boxes_xywh(_t(masks)).cpu().numpy()
    ref_boxes = OF.masks_to_boxes(masks)
    ref_boxes[:, 2] -= ref_boxes[:, 0]; ref_boxes[:, 3] -= ref_boxes[:, 1]
    assert np.array_equal(boxes, ref_boxes)                                       # integer boxes: bit-exact
    got = SU.segmap2segimg(_t(masks), _t(img), also_bbox, bbox_margin=50, out_l=96).cpu().numpy()
    ref = OF.mask_crops(masks, img, also_bbox, 50, 96)
    assert got.shape == ref.shape == (masks.shape[0], 6 if also_bbox else 3, 96, 96)
    d = np.abs(got - ref)
    if as_u8:        # rounded outputs: equal except where the f32 sum lands within an ulp of .5
        assert d.max() <= 1.0 and (d > 0).mean() < 2e-3
    else:
        assert d.max() < 2e-3                                                     # 0..255 scale, f32 accumulation order only
    assert np.all(got[-1, :3] == 0) and np.all(got[-4, :3] == 0)                  # empty / degenerate masked crops -> zeros


@pytest.mark.parametrize("mode,card", [("vanilla", "tiny-clip"), ("fixed_weights", "tiny-clip"), ("hovsg", "tiny-clip"),
                                       ("adaptive_weights", "tiny-clip"), ("concept_fusion", "tiny-clip"),
                                       ("vanilla", "tiny-siglip"), ("hovsg", "tiny-siglip")])
def test_extract_clip_crop_modes_vs_oracle(mode, card):
    """clip_generator.py:125-158 for the crop-based embed types: crops -> ViT pooled descriptors -> fusion; with a CLIP tower
    (class token + proj) and a SigLIP one (attention-pool head; the reference's default model_card is SigLIP-384)."""
    from oracle import features as OF, vit as OV
    from ovo_amd.encoders.vit import SPECS, HipViT, random_state
    from ovo_amd.entities.clip_generator import CLIPGenerator
    spec = SPECS[card]
    sd = random_state(spec, seed=6)
    gen = CLIPGenerator({"embed_type": mode, "model_card": card, "mask_res": 80}, device=DEV, encoder=HipViT(spec, sd, device=DEV))
    masks, img = _crop_fixture()
    masks = masks[:-4]                                   # regular masks only (the reference raises on degenerate boxes)
    got = gen.extract_clip(_t(img), _t(masks)).cpu().numpy()

    def encode(x01):                                     # encode_image: preprocess to the model size, pooled + projected
        b = torch.stack([OV.open_clip_preprocess(torch.from_numpy(np.ascontiguousarray(im)), spec.image_size, spec.mean, spec.std, spec.resize_mode,
                                                 spec.interpolation) for im in x01])
        if spec.map_pool:
            t = OV.vit_forward(sd, b, patch=spec.patch, heads=spec.heads, act=spec.act, pre_ln=False, cls_token=False, eps=spec.ln_eps, tokens=True)
            f = OV.map_pool(sd, t, spec.heads, act=spec.act, eps=spec.ln_eps).numpy()
        else:
            f = OV.vit_forward(sd, b, patch=spec.patch, heads=spec.heads, act=spec.act, rope=None, tokens=False).numpy()
        return f / np.linalg.norm(f, axis=-1, keepdims=True)
    crops = OF.mask_crops(masks, img, mode != "vanilla", 50, 80) / 255.0
    if mode == "vanilla":
        ref = encode(crops[:, :3])
    else:
        g = encode(img[None].astype(np.float32) / 255.0)
        ref = OF.fuse_crop_descriptors(np.repeat(g, len(masks), 0), encode(crops[:, :3]), encode(crops[:, 3:]), mode, gen.w_masked, gen.w_global)
    assert got.shape == ref.shape == (len(masks), spec.out_dim)
    err = np.abs(got - ref).max()
    print(f"{mode}: max |unit descriptor error| = {err:.2e}")
    assert err < _bound(spec.out_dim)               # bf16 encoder vs fp32 oracle, as in test_gpu_encoder
    np.testing.assert_allclose(np.linalg.norm(got, axis=1), 1.0, atol=1e-5)
