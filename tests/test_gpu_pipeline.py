"""-m gpu: the keyframe pipeline (ovo_amd/pipeline.py) -- stream overlap must not change a single bit of the results."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _run(prefetch: bool, side_stream: bool, pipelined: bool = False, n_frames: int = 4, encoder_batch: int = 1):
    from ovo_amd.pipeline import FramePipeline, synthetic_frames
    pipe = FramePipeline(DEV, vit_card="tiny-pe", sam_card="hiera_test", n_map=60_000, n_text=7, scale=0.35, extra_capacity=200_000,
                         track_th=40, encoder_batch=encoder_batch)
    pipe.prefetch = prefetch
    pipe.join_each_step = not pipelined
    if not side_stream:
        pipe.sam_stream = None
    frames = synthetic_frames(n_frames, DEV, scale=0.35, n_masks_grid=(3, 4), n_blobs=4)
    trace = []
    for i, f in enumerate(frames):
        out = pipe.step(f, frames[i + 1:])
        if not pipelined:
            torch.cuda.synchronize()
        with torch.cuda.stream(pipe.sam_stream or torch.cuda.current_stream()):      # the SAM2 features live in a reused workspace
            sam = [t.clone() for t in pipe.sam_frame]                                   # this frame's slice of the (batched) forward
        desc = pipe.ovo.last_clip_embeds
        trace.append({"n_points": out["n_points"], "n_instances": out["n_instances"], "cls": out.get("cls"), "sim": out.get("sim"),
                      "dense_cls": out["dense_cls"].clone(), "dense_conf": out["dense_conf"].clone(), "desc": desc, "sam": sam})   # (the dense map is resident: copy this frame's state)
    torch.cuda.synchronize()                                     # pipelined: the only sync of the run
    for r in trace:
        for k in ("cls", "sim", "dense_cls", "dense_conf", "desc"):
            r[k] = None if r[k] is None else r[k].cpu().numpy()
        r["sam"] = [t.float().cpu().numpy() for t in r["sam"]]
    return trace, pipe


def test_streams_do_not_change_results():
    """SAM2 encoder on its own stream + ViT tokens prefetched on a third: identical bits to the single-stream order."""
    ref, _ = _run(prefetch=False, side_stream=False)
    _check(ref, *_run(prefetch=True, side_stream=True))
    _check(ref, *_run(prefetch=True, side_stream=True, pipelined=True))     # no join at frame boundaries


def _check(ref, got, pipe):
    assert pipe.ovo._vit_stream is not None, "the prefetch path did not run"
    assert any(r["desc"] is not None and r["desc"].shape[0] > 0 for r in ref), "fixture produced no descriptors"
    for a, b in zip(ref, got):
        assert a["n_points"] == b["n_points"] and a["n_instances"] == b["n_instances"]
        for k in ("cls", "sim", "dense_cls", "dense_conf", "desc"):
            if a[k] is None:
                assert b[k] is None
            else:
                assert np.array_equal(a[k], b[k], equal_nan=True), k
        for x, y in zip(a["sam"], b["sam"]):
            assert np.array_equal(x, y)


def test_prefetch_falls_back_for_other_images():
    """A prefetched image that is not the one later pooled must be ignored (kf_queue_delay > 0, dropped keyframes)."""
    from ovo_amd.pipeline import FramePipeline, synthetic_frames
    pipe = FramePipeline(DEV, vit_card="tiny-pe", sam_card=None, n_map=60_000, n_text=3, scale=0.35, extra_capacity=200_000, track_th=40,
                         dense=False)
    f0, f1 = synthetic_frames(2, DEV, scale=0.35, n_masks_grid=(3, 4), n_blobs=4)
    assert pipe.ovo.prefetch_image_features(f1.rgb)                  # tokens of ANOTHER frame are pending ...
    d_other = pipe.ovo._extract_clip(f0.rgb, f0.masks[:5])           # ... and must not be used for f0
    d_plain = pipe.ovo._extract_clip(f0.rgb, f0.masks[:5])
    assert torch.equal(d_other, d_plain)
    assert pipe.ovo.prefetch_image_features(f0.rgb)
    assert torch.equal(pipe.ovo._extract_clip(f0.rgb, f0.masks[:5]), d_plain)


def test_encoder_lookahead_batching_does_not_change_results():
    """Several keyframes' crops per ViT forward and several frames per SAM2 forward (encoder_batch = 3, one group of look-ahead): every
    descriptor, class, dense map and SAM2 feature is bit-identical to the frame-by-frame run -- a row of a GEMM / attention / LayerNorm
    does not depend on how many other rows the launch carries (same k-order in every tile shape)."""
    ref, _ = _run(prefetch=True, side_stream=True, n_frames=7)
    got, pipe = _run(prefetch=True, side_stream=True, n_frames=7, encoder_batch=3, pipelined=True)
    assert pipe.ovo._batch_slots is not None, "the batched prefetch did not run"
    _check(ref, got, pipe)


def test_shared_identical_crops_do_not_change_results():
    """`share_crops` (opt-in): frames so small that the TextRegion tiling is [whole, whole] -- the look-ahead forward then carries ONE crop per
    keyframe instead of two, and every descriptor / class is the same bits as with the reference's duplicate forward."""
    from ovo_amd.pipeline import FramePipeline, synthetic_frames
    frames = synthetic_frames(6, DEV, scale=0.25, n_masks_grid=(3, 4), n_blobs=4)
    runs = []
    for share in (False, True):
        pipe = FramePipeline(DEV, vit_card="tiny-pe", sam_card=None, n_map=60_000, n_text=7, scale=0.25, extra_capacity=200_000, track_th=40,
                             encoder_batch=2, share_crops=share)
        tr = pipe.clip.textregion
        assert len(tr._crops(*frames[0].rgb.shape[:2])) == 2 and len(tr.forward_crops(*frames[0].rgb.shape[:2])) == (1 if share else 2)
        trace = []
        for i, f in enumerate(frames):
            out = pipe.step(f, frames[i + 1:])
            trace.append((out["n_instances"], None if out.get("sim") is None else out["sim"].clone(), pipe.ovo.last_clip_embeds, out["dense_cls"].clone()))
        torch.cuda.synchronize()
        runs.append(trace)
    assert any(d is not None and d.shape[0] > 0 for _, _, d, _ in runs[0]), "fixture produced no descriptors"
    for (na, sa, da, ca), (nb, sb, db, cb) in zip(*runs):
        assert na == nb and torch.equal(ca, cb)
        assert (sa is None and sb is None) or torch.equal(sa, sb)
        assert (da is None and db is None) or torch.equal(torch.nan_to_num(da, nan=7.0), torch.nan_to_num(db, nan=7.0))


@pytest.mark.parametrize("fused", [True, False])
def test_incremental_dense_query_equals_full_requery(fused, monkeypatch):
    """The resident dense class / confidence map, patched only for the rows a keyframe changed -- in one launch from the tracking pass's hit list
    (`ovo_scatter_accum_query`, round 6) or as ovo_scatter_accum_touched -> ovo_similarity_rows --, against a full re-query of every row after
    every keyframe: equal bit for bit, and the touched set is a small part of the map.  The hit list itself = the points with point_seg >= 0."""
    from ovo_amd.pipeline import FramePipeline, synthetic_frames
    from ovo_amd.utils import clip_utils
    if not fused:
        monkeypatch.setenv("OVO_NO_FUSED_SCATTER", "1")
    pipe = FramePipeline(DEV, vit_card="tiny-pe", sam_card=None, n_map=60_000, n_text=7, scale=0.35, extra_capacity=200_000, track_th=40)
    assert pipe.incremental_query and (pipe.ovo.hit_shard is not None) == fused
    frames = synthetic_frames(5, DEV, scale=0.35, n_masks_grid=(3, 4), n_blobs=4)
    fractions, accs = [], []
    for f in frames:
        out = pipe.step(f)
        n = out["n_points"]
        _, cls, conf = clip_utils.similarity(pipe.acc[:n], pipe.texts, cnt=pipe.cnt[:n], want_sim=False, want_argmax=True)
        assert torch.equal(out["dense_cls"], cls) and torch.equal(out["dense_conf"], conf)
        matched = torch.nonzero(pipe.ovo.last_point_seg >= 0).flatten()
        if fused:
            hits = pipe.ovo.last_hits
            touched = int(hits[-4].item())
            assert torch.equal(torch.sort(hits[:touched].long()).values, matched)          # every matched point, once, in any order
        else:
            touched = int(pipe.n_touched[pipe._touch_parity ^ 1].item())
            assert touched == matched.numel() or touched <= n                               # every matched point of a kept mask, once
        fractions.append(touched / n)
        accs.append((pipe.acc[:n].clone(), pipe.cnt[:n].clone()))
    assert (out["dense_cls"] >= 0).any() and max(fractions) > 0 and max(fractions) < 0.6
    return accs


def test_fused_scatter_query_accumulators_equal_the_three_launch_path(monkeypatch):
    """Same frames through both forms: the dense accumulators and counts are identical after every keyframe."""
    a = test_incremental_dense_query_equals_full_requery(True, monkeypatch)
    b = test_incremental_dense_query_equals_full_requery(False, monkeypatch)
    for (xa, ca), (xb, cb) in zip(a, b):
        assert torch.equal(xa, xb) and torch.equal(ca, cb)
    assert float(a[-1][0].abs().sum()) > 0


def test_hit_list_of_a_point_shard():
    """hit_shard_count = 3: the tracking pass lists only the points of this rank's block-cyclic shard, as local rows (`emulate` = rank 1 of 3)."""
    from ovo_amd.pipeline import FramePipeline, synthetic_frames
    pipe = FramePipeline(DEV, vit_card="tiny-pe", sam_card=None, n_map=60_000, n_text=7, scale=0.35, extra_capacity=200_000, track_th=40, emulate=(1, 3))
    assert pipe.ovo.hit_shard == (1, 3, pipe.SHARD_BLOCK)
    frames = synthetic_frames(6, DEV, scale=0.35, n_masks_grid=(3, 4), n_blobs=4)
    seen = 0
    for r in range(2):
        pipe.step_round(frames[3 * r:3 * r + 3], frames[3 * r + 3:])
        seg, hits = pipe.ovo.last_point_seg, pipe.ovo.last_hits               # the round's LAST keyframe
        i = torch.nonzero(seg >= 0).flatten()
        blk = i // pipe.SHARD_BLOCK
        mine = i[blk % 3 == 1]
        local = (blk[blk % 3 == 1] // 3) * pipe.SHARD_BLOCK + mine % pipe.SHARD_BLOCK
        n = int(hits[-4].item())
        assert torch.equal(torch.sort(hits[:n].long()).values, torch.sort(local).values)
        seen += n
    assert seen > 0


def test_keyframe_without_descriptors_releases_its_lookahead_slot():
    """A keyframe whose plan is empty (no mask tracked) never pools its prefetched tokens: the look-ahead slot must still be released,
    or the forward two groups later refuses to overwrite it (encoder_batch = 2, blank keyframes in two different groups)."""
    from ovo_amd.pipeline import Frame, FramePipeline, synthetic_frames
    pipe = FramePipeline(DEV, vit_card="tiny-pe", sam_card=None, n_map=60_000, n_text=3, scale=0.35, extra_capacity=300_000, track_th=40,
                         dense=False, encoder_batch=2)
    frames = synthetic_frames(8, DEV, scale=0.35, n_masks_grid=(3, 4), n_blobs=4)
    for k in (1, 4):                                                 # nothing labelled: no vote, no instance, no descriptor
        f = frames[k]
        frames[k] = Frame(f.index, f.rgb, f.rgb_lr, f.depth, f.c2w, torch.full_like(f.seg_map, -1), torch.zeros_like(f.masks))
    outs = [pipe.step(f, frames[i + 1:]) for i, f in enumerate(frames)]
    torch.cuda.synchronize()
    assert not pipe.ovo._prefetched_batch                            # every prefetched image was consumed or discarded
    assert all(s["left"] == 0 for s in pipe.ovo._batch_slots)
    assert outs[-1]["n_instances"] > 0


def _rounds(extra_capacity, n_rounds=4, world=2, drain_at=None):
    from ovo_amd.pipeline import FramePipeline, synthetic_frames
    pipe = FramePipeline(DEV, vit_card="tiny-pe", sam_card="hiera_test", n_map=60_000, n_text=7, scale=0.35, extra_capacity=extra_capacity,
                         track_th=40, dense=False, emulate=(0, world))
    frames = synthetic_frames(n_rounds * world, DEV, scale=0.35, n_masks_grid=(3, 4), n_blobs=4)
    sizes = []
    for r in range(n_rounds if drain_at is None else drain_at):
        out = pipe.step_round(frames[r * world:(r + 1) * world], frames[(r + 1) * world:])
        sizes.append(out["n_points"])
    if drain_at is not None:
        pipe.drain()
    torch.cuda.synchronize()
    return pipe, sizes


def test_round_that_outgrows_the_map_reservation():
    """A round whose keyframes do not fit the reserved capacity: the map grows ONCE, before the round's first deferred step is
    built (`VanillaMapper.reserve_round`) -- not in the middle of a round whose earlier steps hold the old buffers' addresses -- and the
    run equals the pre-reserved one.  `n_points` of a round comes from the round's last map step."""
    big, sizes_big = _rounds(400_000)
    small, sizes_small = _rounds(0)                                # 60 000 points reserved: the first round already outgrows it
    assert small.slam._cap > 65_536 and sizes_small == sizes_big and sizes_big == sorted(sizes_big) and sizes_big[-1] > 60_000
    assert sizes_big[-1] == big.slam._n == small.slam._n
    for a, b in ((big.slam.pcd, small.slam.pcd), (big.slam.pcd_ids, small.slam.pcd_ids), (big.slam.pcd_obj_ids, small.slam.pcd_obj_ids)):
        assert torch.equal(a, b)
    assert list(big.ovo.objects) == list(small.ovo.objects) and big.ovo.next_ins_id == small.ovo.next_ins_id


def test_drain_reads_back_a_prequeued_round():
    """The next round is queued before this one is read (software pipelining); a stream that ends there calls `drain()`: the host's
    instances then match what the device already did to the map."""
    pipe, _ = _rounds(400_000, n_rounds=3, drain_at=2)             # two rounds stepped, the third pre-queued
    full, _ = _rounds(400_000, n_rounds=3)
    assert not pipe._chains and not pipe.ovo._track_pending
    assert torch.equal(pipe.slam.pcd_obj_ids, full.slam.pcd_obj_ids) and pipe.slam._n == full.slam._n
    assert list(pipe.ovo.objects) == list(full.ovo.objects) and pipe.ovo.next_ins_id == full.ovo.next_ins_id


def _own_run(frames, **kw):
    from ovo_amd.pipeline import FramePipeline
    pipe = FramePipeline(DEV, vit_card="tiny-pe", sam_card="hiera_test256", n_map=60_000, n_text=7, scale=0.35, extra_capacity=200_000, track_th=40, **kw)
    pipe.join_each_step = True
    trace, used = [], []
    for i, f in enumerate(frames):
        out = pipe.step(f, frames[i + 1:])
        torch.cuda.synchronize()
        cur = pipe.masks.frames[f.index]                            # the Frame the round tracked with (its masks replaced in own-mask mode)
        used.append((cur.seg_map.clone(), cur.masks.clone()))
        trace.append((out["n_points"], out["n_instances"], out["dense_cls"].cpu().numpy().copy(), pipe.ovo.last_clip_embeds.cpu().numpy().copy()))
    return trace, used, pipe


def test_own_sam2_masks_drive_the_round():
    """`own_masks` (bench.py --sam-own-masks): a keyframe is tracked with what ITS generator produced -- generate -> mask NMS (masks_update) ->
    mask2segmap, the reference's default chain (mask_generator.py:102-120) -- not with the masks the frame carries.
    (1) the masks in use differ from the carried ones and their count is the generator's; (2) replaying the frames with exactly those masks
    through the precomputed-mask seam reproduces the run bit for bit (the own-mask path adds nothing but the masks); (3) a keyframe whose
    generator keeps fewer than `min_own_masks` masks falls back to the carried masks, after the generator ran, and is counted."""
    from ovo_amd.pipeline import Frame, synthetic_frames
    frames = synthetic_frames(3, DEV, scale=0.35, n_masks_grid=(3, 4), n_blobs=4)
    own, used, pipe = _own_run(frames, sam_full=True, points_per_side=6, own_masks=True, amg_thresholds=(0.0, 0.0), nms_score_thr=0.0)
    assert pipe.own_masks and pipe.own_fallbacks == 0 and len(pipe.own_mask_counts) == len(frames)
    assert all(c > 0 for c in pipe.own_mask_counts)
    for f, (seg, masks), c in zip(frames, used, pipe.own_mask_counts):
        assert masks.shape[0] == c and masks.shape[1:] == f.masks.shape[1:]
        assert masks.shape != f.masks.shape or not torch.equal(masks.view(torch.uint8), f.masks.view(torch.uint8))
        assert int(seg.max()) < c and int(seg.min()) >= -1
    replay = [Frame(f.index, f.rgb, f.rgb_lr, f.depth, f.c2w, seg, masks.view(torch.bool)) for f, (seg, masks) in zip(frames, used)]
    seam, _, _ = _own_run(replay)
    for a, b in zip(own, seam):
        assert a[0] == b[0] and a[1] == b[1]
        assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    # fallback: the generator's masks are too few -> the carried masks are tracked, as without own_masks at all
    fb, used_fb, pipe_fb = _own_run(frames, sam_full=True, points_per_side=6, own_masks=True, amg_thresholds=(0.0, 0.0), nms_score_thr=0.0, min_own_masks=10 ** 6)
    assert pipe_fb.own_fallbacks == len(frames) and pipe_fb.own_mask_counts == pipe.own_mask_counts
    plain, _, _ = _own_run(frames)
    for a, b, (seg, masks), f in zip(fb, plain, used_fb, frames):
        assert torch.equal(masks, f.masks) and torch.equal(seg, f.seg_map)
        assert a[0] == b[0] and a[1] == b[1] and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
