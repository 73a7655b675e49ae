"""HuggingFace Sam2PromptEncoder + Sam2MaskDecoder golden vectors (random weights, small config) for oracle/sam2_decoder.py.

    python tools/gen_hf_sam2_decoder.py        ->  tests/golden/hf_sam2_decoder.npz
"""
import os

import numpy as np
import torch


def main(out_dir):
    from transformers.models.sam2 import modeling_sam2 as M
    from transformers import Sam2MaskDecoderConfig, Sam2PromptEncoderConfig
    torch.manual_seed(0)
    C, S, IMG = 64, 8, 128
    pcfg = Sam2PromptEncoderConfig(hidden_size=C, image_size=IMG, patch_size=IMG // S, num_point_embeddings=4, scale=1)
    dcfg = Sam2MaskDecoderConfig(hidden_size=C, mlp_dim=128, num_hidden_layers=2, num_attention_heads=4, attention_downsample_rate=2,
                                 num_multimask_outputs=3, iou_head_depth=3, iou_head_hidden_dim=C)
    pe, dec = M.Sam2PromptEncoder(pcfg).eval(), M.Sam2MaskDecoder(dcfg).eval()
    with torch.no_grad():
        for p in list(pe.parameters()) + list(dec.parameters()):
            p.add_(torch.randn_like(p) * 0.05)
        P = 5
        points = torch.rand(1, P, 1, 2) * (IMG - 1)                       # [batch, point_batch, points per prompt, xy]
        labels = torch.ones(1, P, 1, dtype=torch.long)
        labels[0, 3, 0] = 0                                                # one background click
        sparse, dense = pe(points, labels, None, None)
        embed = torch.randn(1, C, S, S)
        feat_s0, feat_s1 = torch.randn(1, C // 8, 4 * S, 4 * S), torch.randn(1, C // 4, 2 * S, 2 * S)
        grid = torch.ones(S, S)
        y, x = (grid.cumsum(0) - 0.5) / S, (grid.cumsum(1) - 0.5) / S
        ipe = pe.shared_embedding(torch.stack([x, y], dim=-1)).permute(2, 0, 1)[None]
        arrays = {"points": points[0, :, :, :].numpy(), "labels": labels[0].numpy(), "embed": embed[0].numpy(),
                  "feat_s0": feat_s0[0].numpy(), "feat_s1": feat_s1[0].numpy(), "sparse": sparse[0].numpy(), "image_pe": ipe[0].numpy(),
                  "heads": np.int64(4), "image_size": np.int64(IMG)}
        for multi in (True, False):
            masks, iou, _, obj = dec(image_embeddings=embed, image_positional_embeddings=ipe, sparse_prompt_embeddings=sparse,
                                     dense_prompt_embeddings=dense, multimask_output=multi, high_resolution_features=[feat_s0, feat_s1])
            tag = "multi" if multi else "single"
            arrays[f"masks_{tag}"], arrays[f"iou_{tag}"], arrays[f"obj_{tag}"] = masks[0].numpy(), iou[0].numpy(), obj[0].numpy()
    for k, v in pe.state_dict().items():
        arrays["w:prompt_encoder." + k] = v.numpy()
    for k, v in dec.state_dict().items():
        arrays["w:mask_decoder." + k] = v.numpy()
    path = os.path.join(out_dir, "hf_sam2_decoder.npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB", {k: v.shape for k, v in arrays.items() if not k.startswith("w:")})


if __name__ == "__main__":
    main(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
