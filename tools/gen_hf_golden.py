#!/usr/bin/env python3
"""Golden vectors for the model-forward oracles from HuggingFace transformers (random weights, seeded).

The reference's own model dependencies (open_clip, perception_models, sam2) are not vendored,
so the ViT / Hiera oracles are pinned against an INDEPENDENT implementation of the same published
architectures instead.  Run in the build container:   python tools/gen_hf_golden.py
Writes tests/golden/hf_clip_vit.npz (and hf_sam2_hiera.npz) -- weights, input and outputs, small shapes.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")


def gen_clip():
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    torch.manual_seed(0)
    cfg = CLIPVisionConfig(hidden_size=64, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                           image_size=48, patch_size=16, projection_dim=32, hidden_act="quick_gelu")
    m = CLIPVisionModelWithProjection(cfg).eval()
    with torch.no_grad():
        for p in m.parameters():                      # HF's default init leaves biases / LN trivial
            p.add_(torch.randn_like(p) * 0.05)
        x = torch.randn(3, 3, 48, 48)
        out = m(pixel_values=x, output_hidden_states=True)
    arrays = {"x": x.numpy(), "image_embeds": out.image_embeds.numpy(), "last_hidden": out.last_hidden_state.numpy(),
              "patch": np.int64(16), "heads": np.int64(4)}
    for k, v in m.state_dict().items():
        arrays["w:" + k] = v.numpy()
    path = os.path.join(OUT, "hf_clip_vit.npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    gen_clip()
    if "--sam2" in sys.argv or True:
        try:
            from gen_hf_sam2 import gen_sam2       # optional second file
            gen_sam2(OUT)
        except ImportError:
            pass
