"""-m gpu: the keyframe pipeline (ovo_amd/pipeline.py) -- stream overlap must not change a single bit of the results."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _run(prefetch: bool, side_stream: bool, pipelined: bool = False, n_frames: int = 4):
    from ovo_amd.pipeline import FramePipeline, synthetic_frames
    pipe = FramePipeline(DEV, vit_card="tiny-pe", sam_card="hiera_test", n_map=60_000, n_text=7, scale=0.35, extra_capacity=200_000,
                         track_th=40)
    pipe.prefetch = prefetch
    pipe.join_each_step = not pipelined
    if not side_stream:
        pipe.sam_stream = None
    frames = synthetic_frames(n_frames, DEV, scale=0.35, n_masks_grid=(3, 4), n_blobs=4)
    trace = []
    for f in frames:
        out = pipe.step(f)
        if not pipelined:
            torch.cuda.synchronize()
        with torch.cuda.stream(pipe.sam_stream or torch.cuda.current_stream()):      # the SAM2 features live in a reused workspace
            sam = [t.clone() for t in pipe.sam_out]
        desc = pipe.ovo.last_clip_embeds
        trace.append({"n_points": out["n_points"], "n_instances": out["n_instances"], "cls": out.get("cls"), "sim": out.get("sim"),
                      "dense_cls": out["dense_cls"], "dense_conf": out["dense_conf"], "desc": desc, "sam": sam})
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
