"""HuggingFace Sam2VisionModel golden vectors (random weights, small config) for oracle/hiera.py."""
import os

import numpy as np
import torch


def gen_sam2(out_dir):
    from transformers import Sam2HieraDetConfig, Sam2VisionConfig, Sam2VisionModel
    torch.manual_seed(0)
    bb = Sam2HieraDetConfig(hidden_size=16, num_attention_heads=1, image_size=[128, 128], blocks_per_stage=[1, 2, 3, 2],
                            embed_dim_per_stage=[16, 32, 64, 128], num_attention_heads_per_stage=[1, 2, 4, 8],
                            window_size_per_stage=[8, 4, 6, 4], global_attention_blocks=[4],
                            window_positional_embedding_background_size=[5, 5])
    cfg = Sam2VisionConfig(backbone_config=bb, backbone_channel_list=[128, 64, 32, 16], fpn_hidden_size=32,
                           backbone_feature_sizes=[[32, 32], [16, 16], [8, 8]])
    m = Sam2VisionModel(cfg).eval()
    with torch.no_grad():
        for p in m.parameters():
            p.add_(torch.randn_like(p) * 0.05)
        x = torch.randn(2, 3, 128, 128)
        out = m(pixel_values=x)
    arrays = {"x": x.numpy(), "stages": np.asarray([1, 2, 3, 2]), "heads": np.asarray([1, 2, 4, 8]),
              "window_spec": np.asarray([8, 4, 6, 4]), "global_blocks": np.asarray([4])}
    for i, f in enumerate(out.fpn_hidden_states):           # fine -> coarse, NCHW
        arrays[f"fpn{i}"] = f.numpy()
    for k, v in m.state_dict().items():
        arrays["w:" + k] = v.numpy()
    path = os.path.join(out_dir, "hf_sam2_hiera.npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")
