"""HuggingFace SiglipVisionModel golden vectors (random weights, small config) for oracle/vit.py's SigLIP path
(no class token, tanh-GELU, attention-pooling head).

    python tools/gen_hf_siglip.py        ->  tests/golden/hf_siglip_vit.npz
"""
import os

import numpy as np
import torch


def main(out_dir):
    from transformers import SiglipVisionConfig, SiglipVisionModel
    torch.manual_seed(0)
    cfg = SiglipVisionConfig(hidden_size=64, intermediate_size=176, num_hidden_layers=2, num_attention_heads=4, image_size=56, patch_size=14,
                             hidden_act="gelu_pytorch_tanh", layer_norm_eps=1e-6)
    m = SiglipVisionModel(cfg).eval()
    with torch.no_grad():
        for p in m.parameters():
            p.add_(torch.randn_like(p) * 0.05)
        x = torch.randn(3, 3, 56, 56)
        out = m(pixel_values=x)
    arrays = {"x": x.numpy(), "tokens": out.last_hidden_state.numpy(), "pooled": out.pooler_output.numpy(), "heads": np.int64(4),
              "patch": np.int64(14)}
    for k, v in m.state_dict().items():
        arrays["w:" + k] = v.numpy()
    path = os.path.join(out_dir, "hf_siglip_vit.npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
