"""Oracle: fp32 CPU restatement of the SAM2 image encoder (Hiera trunk + FPN neck), torch functional ops.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED BY THE REFERENCE: the reference reaches this
arithmetic through the un-vendored `sam2` package (segment_utils.py:291-308).  This file restates the published
architecture (SAM 2 paper, Hiera) and is pinned against an independent implementation --
HuggingFace transformers' Sam2VisionModel with random weights, tests/golden/hf_sam2_hiera.npz.

State-dict names follow the sam2 repository (`trunk.*`, `neck.*`, `sam_mask_decoder.conv_s0/1`).
"""
from __future__ import annotations

import math
from typing import Dict, Sequence, Tuple

import torch
import torch.nn.functional as F


def _partition(x: torch.Tensor, ws: int):
    b, h, w, c = x.shape
    ph, pw = (-h) % ws, (-w) % ws
    x = F.pad(x, (0, 0, 0, pw, 0, ph))
    hp, wp = h + ph, w + pw
    x = x.view(b, hp // ws, ws, wp // ws, ws, c).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws, ws, c)
    return x, (hp, wp)


def _unpartition(win: torch.Tensor, ws: int, padded: Tuple[int, int], hw: Tuple[int, int]):
    hp, wp = padded
    b = win.shape[0] // ((hp // ws) * (wp // ws))
    x = win.view(b, hp // ws, wp // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(b, hp, wp, -1)
    return x[:, :hw[0], :hw[1]]


def _pool(x: torch.Tensor) -> torch.Tensor:        # NHWC 2x2 max pool
    return F.max_pool2d(x.permute(0, 3, 1, 2), 2, 2).permute(0, 2, 3, 1)


def hiera_forward(sd: Dict[str, torch.Tensor], images: torch.Tensor, *, stages: Sequence[int], heads: Sequence[int],
                  window_spec: Sequence[int], global_blocks: Sequence[int], eps: float = 1e-6, hi_res: bool = True):
    """images f32 [B, 3, S, S] -> (feat0, feat1, feat2) NHWC, finest first (see ovo_hip.h ovo_hiera_forward)."""
    sd = {k: v.float() for k, v in sd.items()}
    x = F.conv2d(images.float(), sd["trunk.patch_embed.proj.weight"], sd["trunk.patch_embed.proj.bias"], stride=4, padding=3)
    x = x.permute(0, 2, 3, 1)
    h, w = x.shape[1:3]
    win = sd["trunk.pos_embed_window"]
    pos = F.interpolate(sd["trunk.pos_embed"], size=(h, w), mode="bicubic")
    pos = pos + win.tile([1, 1, h // win.shape[2], w // win.shape[3]])
    x = x + pos.permute(0, 2, 3, 1)
    outs, idx = [], 0
    for s, nb in enumerate(stages):
        for b in range(nb):
            p = f"trunk.blocks.{idx}."
            first = s > 0 and b == 0
            ws = window_spec[s - 1] if first else window_spec[s]
            if idx in global_blocks:
                ws = 0
            nh = heads[s]
            hN = F.layer_norm(x, (x.shape[-1],), sd[p + "norm1.weight"], sd[p + "norm1.bias"], eps)
            skip = x
            if p + "proj.weight" in sd:
                skip = _pool(hN @ sd[p + "proj.weight"].T + sd[p + "proj.bias"])
            H, W = hN.shape[1:3]
            t = hN
            if ws > 0:
                t, padded = _partition(t, ws)
            bw, wh, ww, _ = t.shape
            qkv = (t @ sd[p + "attn.qkv.weight"].T + sd[p + "attn.qkv.bias"]).reshape(bw, wh * ww, 3, nh, -1)
            q, k, v = qkv.unbind(2)
            if first:
                q = _pool(q.reshape(bw, wh, ww, -1))
                wh, ww = q.shape[1:3]
                q = q.reshape(bw, wh * ww, nh, -1)
            q, k, v = (z.transpose(1, 2) for z in (q, k, v))
            att = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(q.shape[-1]), dim=-1) @ v
            att = att.transpose(1, 2).reshape(bw, wh, ww, -1) @ sd[p + "attn.proj.weight"].T + sd[p + "attn.proj.bias"]
            if ws > 0:
                wso = ws // 2 if first else ws
                Ho, Wo = skip.shape[1:3]
                att = _unpartition(att, wso, (Ho + (-Ho) % wso, Wo + (-Wo) % wso), (Ho, Wo))
            x = skip + att
            hN = F.layer_norm(x, (x.shape[-1],), sd[p + "norm2.weight"], sd[p + "norm2.bias"], eps)
            u = F.gelu(hN @ sd[p + "mlp.layers.0.weight"].T + sd[p + "mlp.layers.0.bias"])
            x = x + (u @ sd[p + "mlp.layers.1.weight"].T + sd[p + "mlp.layers.1.bias"])
            idx += 1
        outs.append(x)
    lat = []
    for s in range(4):                                   # level s (fine -> coarse) uses neck.convs[3 - s]
        wgt = sd[f"neck.convs.{3 - s}.conv.weight"].reshape(-1, outs[s].shape[-1])
        lat.append(outs[s] @ wgt.T + sd[f"neck.convs.{3 - s}.conv.bias"])
    up = F.interpolate(lat[3].permute(0, 3, 1, 2), scale_factor=2.0, mode="nearest").permute(0, 2, 3, 1)
    f2 = lat[2] + up
    f0, f1 = lat[0], lat[1]
    if hi_res:
        f0 = f0 @ sd["sam_mask_decoder.conv_s0.weight"].reshape(32, -1).T + sd["sam_mask_decoder.conv_s0.bias"]
        f1 = f1 @ sd["sam_mask_decoder.conv_s1.weight"].reshape(64, -1).T + sd["sam_mask_decoder.conv_s1.bias"]
    return f0, f1, f2


def hf_sam2_to_sam2(hf: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Rename a transformers Sam2VisionModel state dict to the sam2 repository's names."""
    out = {"trunk.patch_embed.proj.weight": hf["backbone.patch_embed.projection.weight"],
           "trunk.patch_embed.proj.bias": hf["backbone.patch_embed.projection.bias"],
           "trunk.pos_embed": hf["backbone.pos_embed"], "trunk.pos_embed_window": hf["backbone.pos_embed_window"]}
    ren = {"layer_norm1": "norm1", "layer_norm2": "norm2", "attn.qkv": "attn.qkv", "attn.proj": "attn.proj",
           "mlp.proj_in": "mlp.layers.0", "mlp.proj_out": "mlp.layers.1", "proj": "proj"}
    for k, v in hf.items():
        if k.startswith("backbone.blocks."):
            _, _, i, rest = k.split(".", 3)
            mod, leaf = rest.rsplit(".", 1)
            out[f"trunk.blocks.{i}.{ren[mod]}.{leaf}"] = v
        elif k.startswith("neck.convs."):
            _, _, j, leaf = k.split(".")
            out[f"neck.convs.{j}.conv.{leaf}"] = v
    return out
