"""HuggingFace SiglipTextModel golden vectors (random weights, small config) for oracle/text.py's SigLIP mode.

    python tools/gen_hf_siglip_text.py        ->  tests/golden/hf_siglip_text.npz
"""
import os

import numpy as np
import torch


def main(out_dir):
    from transformers import SiglipTextConfig, SiglipTextModel
    torch.manual_seed(0)
    cfg = SiglipTextConfig(vocab_size=100, hidden_size=64, intermediate_size=176, num_hidden_layers=3, num_attention_heads=4,
                           max_position_embeddings=16, hidden_act="gelu_pytorch_tanh", layer_norm_eps=1e-6, projection_size=64)
    m = SiglipTextModel(cfg).eval()
    arrays = {}
    with torch.no_grad():
        for p in m.parameters():
            p.add_(torch.randn_like(p) * 0.05)
        ids = torch.randint(2, 100, (5, 16))
        for r, n in enumerate((3, 7, 16, 10, 5)):              # SigLIP pads to the context length: padding id 1 after the text
            ids[r, n:] = 1
        out = m(input_ids=ids).pooler_output
    arrays["ids"], arrays["out"], arrays["heads"] = ids.numpy(), out.numpy(), np.int64(4)
    for k, v in m.state_dict().items():
        arrays[f"w:{k}"] = v.numpy()
    path = os.path.join(out_dir, "hf_siglip_text.npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", sorted(k for k in arrays if k.startswith("w:"))[:6])


if __name__ == "__main__":
    main(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
