"""-m gpu: SAM2 prompt encoder + mask decoder (f1) on MI355X vs the fp32 oracle (pinned to HuggingFace's implementation)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _inputs(spec, seed=0):
    g = torch.Generator().manual_seed(seed)
    s, c = spec.embed_size, spec.hidden
    emb = torch.randn(s, s, c, generator=g)
    f1 = torch.randn(2 * s, 2 * s, c // 4, generator=g) * 0.5
    f0 = torch.randn(4 * s, 4 * s, c // 8, generator=g) * 0.5
    return emb, f1, f0


@pytest.mark.parametrize("card,n_side", [("sam2_test", 3), ("sam2", 2)])
def test_mask_decoder_vs_oracle(card, n_side):
    from oracle import sam2_decoder as SD
    from ovo_amd.encoders.sam_decoder import SPECS, HipSamDecoder, random_state
    spec = SPECS[card]
    sd = random_state(spec, seed=2)
    dec = HipSamDecoder(spec, sd, device=DEV)
    pts = dec.set_point_grid(n_side)
    emb, f1, f0 = _inputs(spec)
    masks, iou = dec.forward(emb.to(DEV), f1.to(DEV), f0.to(DEV))
    # oracle: NCHW inputs, the same prompt tokens built by its own point embedding
    labels = torch.ones(pts.shape[0], 1, dtype=torch.long)
    sparse = SD.embed_points(sd, pts.float()[:, None, :], labels, spec.image_size)
    np.testing.assert_allclose(dec.tokens0.reshape(pts.shape[0], -1, spec.hidden)[:, 6:].cpu().numpy(), sparse.numpy(), atol=2e-5, rtol=0)
    rm, ri, _ = SD.mask_decoder(sd, emb.permute(2, 0, 1), f1.permute(2, 0, 1), f0.permute(2, 0, 1), sparse, heads=spec.heads, multimask=True)
    assert masks.shape == rm.shape and iou.shape == ri.shape
    m, r = masks.cpu(), rm
    rms = r.pow(2).mean().sqrt().item()
    err = (m - r).abs().max().item()
    cos = torch.nn.functional.cosine_similarity(m.flatten(1), r.flatten(1), dim=1).min().item()
    agree = ((m > 0) == (r > 0)).float().mean().item()
    print(f"{card}: mask logits max err / rms = {err / rms:.3e}, min cosine = {cos:.6f}, sign agreement = {agree:.5f}, "
          f"iou max err = {(iou.cpu() - ri).abs().max().item():.2e}")
    # bf16 GEMM operands through 2 two-way layers + 2 upscaling stages, fp32 accumulation / residuals / LayerNorm
    assert err / rms < 0.08 and cos > 0.9995 and agree > 0.995
    assert (iou.cpu() - ri).abs().max().item() < 2e-2
