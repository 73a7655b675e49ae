"""Oracle: fp32 CPU restatement of the CLIP text tower (open_clip `TextTransformer` / `encode_text`), torch functional ops.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED BY THE REFERENCE: the reference calls
`self.model.encode_text(self.tokenizer(text_list))` of the un-vendored open_clip / perception_models packages
(clip_generator.py:161-173).  This file restates the published architecture (CLIP paper: token + learned position
embeddings, pre-LN transformer with a causal mask, final LayerNorm, features of the end-of-text token -- the highest
token id -- times `text_projection`) and is pinned against an independent implementation, HuggingFace transformers'
CLIPTextModelWithProjection with random weights (tests/golden/hf_clip_text.npz).

State-dict names follow open_clip (`token_embedding.weight`, `positional_embedding`, `transformer.resblocks.*`,
`ln_final.*`, `text_projection`).
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F


def text_forward(sd: Dict[str, torch.Tensor], tokens: torch.Tensor, heads: int, act: str = "gelu", eps: float = 1e-5,
                 causal: bool = True, pool: str = "argmax") -> torch.Tensor:
    """tokens i64 [B, T] -> f32 [B, embed] (NOT normalised: clip_generator.py:171-172 normalises afterwards).
    SigLIP text towers (open_clip text_cfg no_causal_mask / pool_type "last" / proj_bias): causal=False, pool="last",
    act="gelu_tanh", eps=1e-6, projection given as `text_projection.weight` [out, w] + `text_projection.bias`; pinned against
    HuggingFace SiglipTextModel (tests/golden/hf_siglip_text.npz)."""
    b, t = tokens.shape
    x = sd["token_embedding.weight"][tokens] + sd["positional_embedding"][:t]
    w = x.shape[-1]
    hd = w // heads
    mask = torch.full((t, t), float("-inf")).triu_(1) if causal else torch.zeros(t, t)
    i = 0
    while f"transformer.resblocks.{i}.ln_1.weight" in sd:
        p = f"transformer.resblocks.{i}."
        h = F.layer_norm(x, (w,), sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], eps)
        qkv = F.linear(h, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"]).view(b, t, 3, heads, hd)
        q, k, v = (qkv[:, :, j].transpose(1, 2) for j in range(3))
        a = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5 + mask, dim=-1) @ v
        x = x + F.linear(a.transpose(1, 2).reshape(b, t, w), sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"])
        h = F.layer_norm(x, (w,), sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], eps)
        h = F.linear(h, sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"])
        h = h * torch.sigmoid(1.702 * h) if act == "quick_gelu" else F.gelu(h, approximate="tanh" if act == "gelu_tanh" else "none")
        x = x + F.linear(h, sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"])
        i += 1
    x = F.layer_norm(x, (w,), sd["ln_final.weight"], sd["ln_final.bias"], eps)
    pooled = x[:, -1] if pool == "last" else x[torch.arange(b), tokens.argmax(dim=-1)]
    if "text_projection.weight" in sd:
        return F.linear(pooled, sd["text_projection.weight"], sd.get("text_projection.bias"))
    return pooled @ sd["text_projection"]


def hf_siglip_text_to_openclip(hf: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """HuggingFace SiglipTextModel parameter names -> open_clip names (projection kept as a Linear with bias)."""
    hf = {(k[len("text_model."):] if k.startswith("text_model.") else k): v for k, v in hf.items()}
    out: Dict[str, torch.Tensor] = {"token_embedding.weight": hf["embeddings.token_embedding.weight"],
                                    "positional_embedding": hf["embeddings.position_embedding.weight"]}
    i = 0
    while f"encoder.layers.{i}.layer_norm1.weight" in hf:
        s, d = f"encoder.layers.{i}.", f"transformer.resblocks.{i}."
        for a, b in (("layer_norm1", "ln_1"), ("layer_norm2", "ln_2"), ("mlp.fc1", "mlp.c_fc"), ("mlp.fc2", "mlp.c_proj"),
                     ("self_attn.out_proj", "attn.out_proj")):
            out[d + b + ".weight"], out[d + b + ".bias"] = hf[s + a + ".weight"], hf[s + a + ".bias"]
        out[d + "attn.in_proj_weight"] = torch.cat([hf[s + f"self_attn.{n}_proj.weight"] for n in "qkv"])
        out[d + "attn.in_proj_bias"] = torch.cat([hf[s + f"self_attn.{n}_proj.bias"] for n in "qkv"])
        i += 1
    out["ln_final.weight"], out["ln_final.bias"] = hf["final_layer_norm.weight"], hf["final_layer_norm.bias"]
    out["text_projection.weight"], out["text_projection.bias"] = hf["head.weight"], hf["head.bias"]
    return out


def hf_clip_text_to_openclip(hf: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """HuggingFace CLIPTextModelWithProjection parameter names -> open_clip names (q/k/v fused into in_proj)."""
    out: Dict[str, torch.Tensor] = {}
    e = "text_model.embeddings."
    out["token_embedding.weight"] = hf[e + "token_embedding.weight"]
    out["positional_embedding"] = hf[e + "position_embedding.weight"]
    i = 0
    while f"text_model.encoder.layers.{i}.layer_norm1.weight" in hf:
        s, d = f"text_model.encoder.layers.{i}.", f"transformer.resblocks.{i}."
        for a, b in (("layer_norm1", "ln_1"), ("layer_norm2", "ln_2"), ("mlp.fc1", "mlp.c_fc"), ("mlp.fc2", "mlp.c_proj"),
                     ("self_attn.out_proj", "attn.out_proj")):
            out[d + b + ".weight"], out[d + b + ".bias"] = hf[s + a + ".weight"], hf[s + a + ".bias"]
        out[d + "attn.in_proj_weight"] = torch.cat([hf[s + f"self_attn.{n}_proj.weight"] for n in "qkv"])
        out[d + "attn.in_proj_bias"] = torch.cat([hf[s + f"self_attn.{n}_proj.bias"] for n in "qkv"])
        i += 1
    out["ln_final.weight"], out["ln_final.bias"] = hf["text_model.final_layer_norm.weight"], hf["text_model.final_layer_norm.bias"]
    out["text_projection"] = hf["text_projection.weight"].t().contiguous()
    return out
