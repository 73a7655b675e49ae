"""HuggingFace CLIPTextModelWithProjection golden vectors (random weights, small config) for oracle/text.py.

    python tools/gen_hf_text.py        ->  tests/golden/hf_clip_text.npz
"""
import os

import numpy as np
import torch


def main(out_dir):
    from transformers import CLIPTextConfig, CLIPTextModelWithProjection
    torch.manual_seed(0)
    arrays = {}
    for act in ("quick_gelu", "gelu"):
        cfg = CLIPTextConfig(vocab_size=100, hidden_size=64, intermediate_size=128, projection_dim=32, num_hidden_layers=3,
                             num_attention_heads=4, max_position_embeddings=16, hidden_act=act, eos_token_id=99, bos_token_id=98, pad_token_id=0)
        m = CLIPTextModelWithProjection(cfg).eval()
        with torch.no_grad():
            for p in m.parameters():
                p.add_(torch.randn_like(p) * 0.05)
            ids = torch.randint(1, 98, (5, 16))
            ids[:, 0] = 98
            for r, n in enumerate((3, 7, 16, 10, 5)):            # end-of-text (= highest id) at different positions, padding after it
                ids[r, n - 1] = 99
                ids[r, n:] = 0
            out = m(input_ids=ids).text_embeds
        arrays[f"{act}:ids"], arrays[f"{act}:out"] = ids.numpy(), out.numpy()
        for k, v in m.state_dict().items():
            arrays[f"{act}:w:{k}"] = v.numpy()
    arrays["heads"] = np.int64(4)
    path = os.path.join(out_dir, "hf_clip_text.npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
