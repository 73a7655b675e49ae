"""Oracle: fp32 CPU restatement of SAM2's prompt encoder (point prompts) and mask decoder, torch functional ops.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED BY THE REFERENCE: the reference reaches this
arithmetic through the un-vendored `sam2` package (`SAM2AutomaticMaskGenerator`, segment_utils.py:291-308,
mask_generator.py:113).  This file restates the published architecture (SAM / SAM 2 papers: random-Fourier point
encoding, two-way transformer, hyper-network mask heads) and is pinned against an independent
implementation -- HuggingFace transformers' Sam2PromptEncoder + Sam2MaskDecoder with random weights,
tests/golden/hf_sam2_decoder.npz.

State-dict names follow the sam2 repository (`sam_prompt_encoder.*`, `sam_mask_decoder.*`).
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch
import torch.nn.functional as F

PE = "sam_prompt_encoder."
MD = "sam_mask_decoder."


def fourier_pe(coords01: torch.Tensor, gauss: torch.Tensor) -> torch.Tensor:
    """coords in [0,1]^2 (x, y), [..., 2] -> [..., 2 * gauss.shape[1]] = [sin | cos] of 2 pi (2c - 1) G."""
    c = (2.0 * coords01 - 1.0) @ gauss
    c = 2.0 * math.pi * c
    return torch.cat([torch.sin(c), torch.cos(c)], dim=-1)


def image_pe(sd: Dict[str, torch.Tensor], size: int) -> torch.Tensor:
    """Dense positional encoding of the size x size embedding grid (pixel centres) -> [size*size, C]."""
    g = sd[PE + "pe_layer.positional_encoding_gaussian_matrix"]
    t = (torch.arange(size, dtype=torch.float32) + 0.5) / size
    yy, xx = torch.meshgrid(t, t, indexing="ij")
    return fourier_pe(torch.stack([xx, yy], dim=-1), g).reshape(size * size, -1)


def embed_points(sd: Dict[str, torch.Tensor], points: torch.Tensor, labels: torch.Tensor, image_size: int) -> torch.Tensor:
    """points [P, n, 2] (x, y) in pixels of the image_size^2 model input, labels [P, n] (1 = foreground, 0 = background)
    -> sparse prompt tokens [P, n + 1, C]: the Fourier code of the pixel CENTRE plus the label embedding, followed by
    the "not a point" padding token SAM appends when there is no box prompt."""
    g = sd[PE + "pe_layer.positional_encoding_gaussian_matrix"]
    pe = fourier_pe((points.float() + 0.5) / float(image_size), g)
    lab = torch.stack([sd[PE + f"point_embeddings.{i}.weight"][0] for i in (0, 1)])        # background, foreground
    pe = pe + lab[labels.long()]
    pad = sd[PE + "not_a_point_embed.weight"][0].expand(points.shape[0], 1, -1)
    return torch.cat([pe, pad], dim=1)


def _attn(sd, pre: str, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, heads: int) -> torch.Tensor:
    """SAM attention (optionally down-projected inner width): [B, Tq, C], [B, Tk, C] -> [B, Tq, C]."""
    q = F.linear(q, sd[pre + "q_proj.weight"], sd[pre + "q_proj.bias"])
    k = F.linear(k, sd[pre + "k_proj.weight"], sd[pre + "k_proj.bias"])
    v = F.linear(v, sd[pre + "v_proj.weight"], sd[pre + "v_proj.bias"])
    b, tq, ci = q.shape
    hd = ci // heads

    def split(x):
        return x.view(b, -1, heads, hd).transpose(1, 2)
    a = torch.softmax(split(q) @ split(k).transpose(-1, -2) * hd ** -0.5, dim=-1) @ split(v)
    return F.linear(a.transpose(1, 2).reshape(b, tq, ci), sd[pre + "out_proj.weight"], sd[pre + "out_proj.bias"])


def _ln(sd, pre: str, x: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), sd[pre + "weight"], sd[pre + "bias"], eps)


def _mlp(sd, pre: str, x: torch.Tensor, n: int, sigmoid: bool = False) -> torch.Tensor:
    for i in range(n):
        x = F.linear(x, sd[pre + f"layers.{i}.weight"], sd[pre + f"layers.{i}.bias"])
        if i < n - 1:
            x = F.relu(x)
    return torch.sigmoid(x) if sigmoid else x


def two_way_transformer(sd, tokens: torch.Tensor, keys: torch.Tensor, key_pe: torch.Tensor, heads: int, depth: int = 2):
    """tokens [P, T, C] (also the query positional code), keys [P, S, C], key_pe [S, C] -> (tokens, keys)."""
    t = MD + "transformer."
    q, q_pe = tokens, tokens
    for i in range(depth):
        L = t + f"layers.{i}."
        if i == 0:                                               # first layer: no positional code, no residual
            q = _attn(sd, L + "self_attn.", q, q, q, heads)
        else:
            q = q + _attn(sd, L + "self_attn.", q + q_pe, q + q_pe, q, heads)
        q = _ln(sd, L + "norm1.", q)
        q = q + _attn(sd, L + "cross_attn_token_to_image.", q + q_pe, keys + key_pe, keys, heads)
        q = _ln(sd, L + "norm2.", q)
        q = q + _mlp(sd, L + "mlp.", q, 2)
        q = _ln(sd, L + "norm3.", q)
        keys = keys + _attn(sd, L + "cross_attn_image_to_token.", keys + key_pe, q + q_pe, q, heads)
        keys = _ln(sd, L + "norm4.", keys)
    q = q + _attn(sd, t + "final_attn_token_to_image.", q + q_pe, keys + key_pe, keys, heads)
    return _ln(sd, t + "norm_final_attn.", q), keys


def mask_decoder(sd: Dict[str, torch.Tensor], image_embed: torch.Tensor, feat_s1: torch.Tensor, feat_s0: torch.Tensor,
                 sparse: torch.Tensor, heads: int = 8, multimask: bool = True) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """image_embed [C, S, S]; feat_s1 [C/4, 2S, 2S], feat_s0 [C/8, 4S, 4S] (conv_s1 / conv_s0 already applied, as the
    image encoder emits them); sparse [P, n, C] prompt tokens.
    -> (mask logits [P, 3 | 4, 4S, 4S], predicted IoU [P, 3 | 4], object-score logits [P, 1])."""
    c, s, _ = image_embed.shape
    p = sparse.shape[0]
    out_tok = torch.cat([sd[MD + "obj_score_token.weight"], sd[MD + "iou_token.weight"], sd[MD + "mask_tokens.weight"]], 0)
    n_mask = sd[MD + "mask_tokens.weight"].shape[0]
    tokens = torch.cat([out_tok[None].expand(p, -1, -1), sparse], dim=1)
    dense = sd[PE + "no_mask_embed.weight"].reshape(c, 1, 1)
    if "no_mem_embed" in sd:                                     # SAM2ImagePredictor.set_image adds it to the coarsest feature [upstream-knowledge]
        dense = dense + sd["no_mem_embed"].reshape(c, 1, 1)
    keys = (image_embed + dense).reshape(c, s * s).t()[None].expand(p, -1, -1)
    q, keys = two_way_transformer(sd, tokens, keys, image_pe(sd, s), heads)
    iou_tok, mask_tok = q[:, 1], q[:, 2:2 + n_mask]
    x = keys.transpose(1, 2).reshape(p, c, s, s)
    up = MD + "output_upscaling."
    x = F.conv_transpose2d(x, sd[up + "0.weight"], sd[up + "0.bias"], stride=2) + feat_s1[None]
    x = x.permute(0, 2, 3, 1)                                    # LayerNorm2d (over channels), eps 1e-6
    x = F.layer_norm(x, (x.shape[-1],), sd[up + "1.weight"], sd[up + "1.bias"], 1e-6).permute(0, 3, 1, 2)
    x = F.gelu(x)
    x = F.gelu(F.conv_transpose2d(x, sd[up + "3.weight"], sd[up + "3.bias"], stride=2) + feat_s0[None])
    hyper = torch.stack([_mlp(sd, MD + f"output_hypernetworks_mlps.{i}.", mask_tok[:, i], 3) for i in range(n_mask)], dim=1)
    masks = (hyper @ x.reshape(p, x.shape[1], -1)).reshape(p, n_mask, 4 * s, 4 * s)
    iou = _mlp(sd, MD + "iou_prediction_head.", iou_tok, 3, sigmoid=True)
    obj = _mlp(sd, MD + "pred_obj_score_head.", q[:, 0], 3)
    if multimask:
        masks, iou = masks[:, 1:], iou[:, 1:]
    return masks, iou, obj


def hf_sam2_decoder_to_sam2(hf: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """HuggingFace Sam2Model parameter names (prompt_encoder.* / mask_decoder.*) -> sam2 repository names."""
    out: Dict[str, torch.Tensor] = {}
    for k, v in hf.items():
        if k == "prompt_encoder.shared_embedding.positional_embedding":
            out[PE + "pe_layer.positional_encoding_gaussian_matrix"] = v
        elif k == "prompt_encoder.point_embed.weight":
            for i in range(v.shape[0]):
                out[PE + f"point_embeddings.{i}.weight"] = v[i:i + 1]
        elif k in ("prompt_encoder.not_a_point_embed.weight", "prompt_encoder.no_mask_embed.weight"):
            out[PE + k.split(".", 1)[1]] = v
        elif k.startswith("mask_decoder."):
            n = k[len("mask_decoder."):]
            n = n.replace(".o_proj.", ".out_proj.")
            for a, b in (("layer_norm1", "norm1"), ("layer_norm2", "norm2"), ("layer_norm3", "norm3"), ("layer_norm4", "norm4"),
                         ("layer_norm_final_attn", "norm_final_attn"), ("upscale_conv1", "output_upscaling.0"),
                         ("upscale_layer_norm", "output_upscaling.1"), ("upscale_conv2", "output_upscaling.3")):
                n = n.replace(a, b)
            if ".proj_in." in n or ".proj_out." in n or (".layers." in n and ("mlps" in n or "head" in n)):
                # Sam2FeedForward(proj_in, layers.*, proj_out) -> MLP.layers.{0..n-1}
                base, leaf = n.rsplit(".", 1)
                if base.endswith(".proj_in"):
                    n = base[:-len(".proj_in")] + ".layers.0." + leaf
                elif base.endswith(".proj_out"):
                    depth = 2 if ".mlp" in base and "mlps" not in base else 3
                    n = base[:-len(".proj_out")] + f".layers.{depth - 1}." + leaf
                else:                                            # hidden layer j -> layers.{j+1}
                    head, j = base.rsplit(".layers.", 1)
                    n = head + f".layers.{int(j) + 1}." + leaf
            out[MD + n] = v
    return out
