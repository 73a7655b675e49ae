"""-m gpu: the keyframe pipeline (ovo_amd/pipeline.py) -- stream overlap must not change a single bit of the results."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _run(prefetch: bool, side_stream: bool, n_frames: int = 4):
    from ovo_amd.pipeline import FramePipeline, synthetic_frames
    pipe = FramePipeline(DEV, vit_card="tiny-pe", sam_card="hiera_test", n_map=60_000, n_text=7, scale=0.35, extra_capacity=200_000,
                         track_th=40)
    pipe.prefetch = prefetch
    if not side_stream:
        pipe.sam_stream = None
    frames = synthetic_frames(n_frames, DEV, scale=0.35, n_masks_grid=(3, 4), n_blobs=4)
    trace = []
    for f in frames:
        out = pipe.step(f)
        torch.cuda.synchronize()
        trace.append({"n_points": out["n_points"], "n_instances": out["n_instances"],
                      "cls": out["cls"].cpu().numpy() if "cls" in out else None,
                      "sim": out["sim"].cpu().numpy() if "sim" in out else None,
                      "dense_cls": out["dense_cls"].cpu().numpy(), "dense_conf": out["dense_conf"].cpu().numpy(),
                      "desc": pipe.ovo.last_clip_embeds.cpu().numpy() if pipe.ovo.last_clip_embeds is not None else None,
                      "sam": [t.float().cpu().numpy() for t in pipe.sam_out]})
    return trace, pipe


def test_streams_do_not_change_results():
    """SAM2 encoder on its own stream + ViT tokens prefetched on a third: identical bits to the single-stream order."""
    ref, _ = _run(prefetch=False, side_stream=False)
    got, pipe = _run(prefetch=True, side_stream=True)
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
