"""CLIP text tower on MI355X.

The reference embeds its class / query prompts with `model.encode_text(tokenizer(text_list))` of open_clip /
perception_models (clip_generator.py:161-173).  Neither package nor a BPE vocabulary file exists offline, so this module
covers the tower itself -- token ids in, embeddings out -- with open_clip's parameter names; a tokenizer (any callable
`list[str] -> i64 [n, context]`) is injected by the caller.  The tower is not on the per-frame path (prompts are embedded
once per query set): it reuses the MFMA GEMM, the fused attention kernel (now with a causal mask, `ovo_attention_t.causal`)
and the row-epilogue kernel, driven from Python.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional

import torch

from .. import _lib as L


@dataclass(frozen=True)
class TextSpec:
    name: str
    vocab: int
    context: int
    width: int
    layers: int
    heads: int
    out_dim: int
    act: str = "gelu"             # "quick_gelu" for the OpenAI / "-quickgelu" open_clip cards; "gelu_tanh" for SigLIP
    mlp_dim: int = 0              # 0 = 4 * width
    causal: bool = True           # SigLIP: text_cfg.no_causal_mask
    pool: str = "argmax"          # "argmax" = features of the end-of-text (highest id) token; "last" = last position (SigLIP)
    proj_bias: bool = False       # SigLIP: the projection is a Linear with bias
    ln_eps: float = 1e-5

    @property
    def hidden(self) -> int:
        return self.mlp_dim or 4 * self.width


# open_clip model configs (text side) of the cards in clip_utils.py:65-75; PE-Core-L14-336: perception_models' text config
SPECS: Dict[str, TextSpec] = {
    "ViT-B-16-qg": TextSpec("ViT-B-16-qg", 49408, 77, 512, 12, 8, 512, "quick_gelu"),
    "ViT-L-14-qg": TextSpec("ViT-L-14-qg", 49408, 77, 768, 12, 12, 768, "quick_gelu"),
    "ViT-H-14": TextSpec("ViT-H-14", 49408, 77, 1024, 24, 16, 1024),
    "PE-Core-L14-336": TextSpec("PE-Core-L14-336", 49408, 32, 1024, 24, 16, 1024),
    # SigLIP so400m text towers (open_clip ViT-SO400M-14-SigLIP[-384] text_cfg; SigLIP2: 256k Gemma vocabulary)
    "SigLIP": TextSpec("SigLIP", 32000, 16, 1152, 27, 16, 1152, "gelu_tanh", 4304, False, "last", True, 1e-6),
    "SigLIP-384": TextSpec("SigLIP-384", 32000, 64, 1152, 27, 16, 1152, "gelu_tanh", 4304, False, "last", True, 1e-6),
    "SigLIP2-384": TextSpec("SigLIP2-384", 256000, 64, 1152, 27, 16, 1152, "gelu_tanh", 4304, False, "last", True, 1e-6),
    "tiny-siglip-text": TextSpec("tiny-siglip-text", 100, 16, 128, 2, 4, 128, "gelu_tanh", 432, False, "last", True, 1e-6),
    "tiny-clip": TextSpec("tiny-clip", 714, 16, 64, 2, 4, 64, "quick_gelu"),       # pairs with the tiny-clip image tower (200-merge test vocabulary)
    "tiny-text": TextSpec("tiny-text", 100, 16, 64, 3, 4, 32, "quick_gelu"),
}


def random_state(spec: TextSpec, seed: int = 0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    w, m = spec.width, spec.hidden
    sd = {"token_embedding.weight": torch.randn(spec.vocab, w, generator=g) * 0.02,
          "positional_embedding": torch.randn(spec.context, w, generator=g) * 0.01,
          "ln_final.weight": 1.0 + 0.1 * torch.randn(w, generator=g), "ln_final.bias": 0.1 * torch.randn(w, generator=g),
          "text_projection": torch.randn(w, spec.out_dim, generator=g) * w ** -0.5}
    for i in range(spec.layers):
        p = f"transformer.resblocks.{i}."
        for n in ("ln_1", "ln_2"):
            sd[p + n + ".weight"], sd[p + n + ".bias"] = 1.0 + 0.1 * torch.randn(w, generator=g), 0.1 * torch.randn(w, generator=g)
        sd[p + "attn.in_proj_weight"] = torch.randn(3 * w, w, generator=g) * w ** -0.5
        sd[p + "attn.in_proj_bias"] = torch.randn(3 * w, generator=g) * 0.02
        sd[p + "attn.out_proj.weight"] = torch.randn(w, w, generator=g) * w ** -0.5
        sd[p + "attn.out_proj.bias"] = torch.randn(w, generator=g) * 0.02
        sd[p + "mlp.c_fc.weight"] = torch.randn(m, w, generator=g) * w ** -0.5
        sd[p + "mlp.c_fc.bias"] = torch.randn(m, generator=g) * 0.02
        sd[p + "mlp.c_proj.weight"] = torch.randn(w, m, generator=g) * m ** -0.5
        sd[p + "mlp.c_proj.bias"] = torch.randn(w, generator=g) * 0.02
    if spec.proj_bias:
        sd["text_projection.weight"], sd["text_projection.bias"] = sd.pop("text_projection").t().contiguous(), torch.randn(spec.out_dim, generator=g) * 0.02
    return sd


class HipTextEncoder:
    def __init__(self, spec: TextSpec, state: Optional[Dict[str, torch.Tensor]] = None, device="cuda", seed: int = 0,
                 tokenizer: Optional[Callable[[List[str]], torch.Tensor]] = None):
        self.spec, self.device, self.tokenizer = spec, torch.device(device), tokenizer
        sd = state if state is not None else random_state(spec, seed)
        if any(k.startswith("text.") for k in sd):                    # open_clip "custom text" checkpoints nest the tower
            sd = {k[len("text."):]: v for k, v in sd.items() if k.startswith("text.")}
        bf, f32 = torch.bfloat16, torch.float32
        dev = self.device
        self.tok = sd["token_embedding.weight"].to(dev, f32).contiguous()
        self.pos = sd["positional_embedding"].to(dev, f32).contiguous()
        self.w: Dict[str, torch.Tensor] = {"ln_final.g": sd["ln_final.weight"].to(dev, f32), "ln_final.b": sd["ln_final.bias"].to(dev, f32)}
        if "text_projection.weight" in sd:                              # nn.Linear form (SigLIP, proj_bias)
            self.w["proj.w"] = sd["text_projection.weight"].to(dev, bf).contiguous()
            if "text_projection.bias" in sd:
                self.w["proj.b"] = sd["text_projection.bias"].to(dev, f32).contiguous()
        else:
            self.w["proj.w"] = sd["text_projection"].t().to(dev, bf).contiguous()
        self.q_prescaled = L.q_prescale_enabled()
        pad = (-spec.hidden) % 32                                        # zero hidden units up to a multiple of 32 (so400m: 4304 -> 4320)
        self.layers = 0
        while f"transformer.resblocks.{self.layers}.ln_1.weight" in sd:
            p = f"transformer.resblocks.{self.layers}."
            for n in ("ln_1", "ln_2"):
                self.w[p + n + ".g"], self.w[p + n + ".b"] = sd[p + n + ".weight"].to(dev, f32), sd[p + n + ".bias"].to(dev, f32)
            for n, src in (("qkv", "attn.in_proj_"), ("out", "attn.out_proj."), ("fc1", "mlp.c_fc."), ("fc2", "mlp.c_proj.")):
                wt, bs = sd[p + src + "weight"].float(), sd[p + src + "bias"].float()
                if n == "qkv" and self.q_prescaled:                      # log2 e / sqrt(hd) into the q rows, in f32, before the bf16 rounding
                    wt, bs = L.fold_q_scale(wt, bs, spec.width, spec.width // spec.heads)
                if pad and n == "fc1":
                    wt, bs = torch.cat([wt, wt.new_zeros(pad, wt.shape[1])]), torch.cat([bs, bs.new_zeros(pad)])
                if pad and n == "fc2":
                    wt = torch.cat([wt, wt.new_zeros(wt.shape[0], pad)], dim=1)
                self.w[p + n + ".w"] = wt.to(dev, bf).contiguous()
                self.w[p + n + ".b"] = bs.to(dev, f32).contiguous()
            self.layers += 1

    def _gemm(self, a, wname, out_dtype, act=0, add=None, out=None, bias=True):
        w = self.w[wname + ".w"]
        m, (n, k) = a.shape[0], w.shape
        if out is None:
            out = torch.empty((m, n), dtype=out_dtype, device=a.device)
        g = L.Gemm()
        g.A, g.lda, g.W, g.ldw = a.data_ptr(), a.stride(0), w.data_ptr(), k
        g.bias = self.w[wname + ".b"].data_ptr() if bias else None
        g.C, g.ldc = out.data_ptr(), out.stride(0)
        g.add, g.ld_add = (add.data_ptr(), add.stride(0)) if add is not None else (None, 0)
        g.M, g.N, g.K = m, n, k
        g.in_dtype, g.out_dtype, g.act, g.alpha = 2, L.DTYPE_CODE[out.dtype], act, 1.0
        L.check(L.load().ovo_gemm(C.byref(g), L.stream()))
        return out

    @torch.no_grad()
    def encode_tokens(self, tokens: torch.Tensor) -> torch.Tensor:
        """tokens i64 [B, T <= context] -> f32 [B, out_dim] (not normalised, like `encode_text`)."""
        spec, lib = self.spec, L.load()
        tokens = tokens.to(self.device).long()
        b, t = tokens.shape
        if t > spec.context:
            raise L.OvoHipError(f"{t} tokens exceed the context length {spec.context}")
        w, H = spec.width, spec.heads
        hd, R = w // H, b * t
        bf, f32 = torch.bfloat16, torch.float32
        x = L.gather_rows(self.tok, tokens.reshape(-1).tolist())                       # [R, w] f32 token embeddings
        h16 = torch.empty((R, w), dtype=bf, device=self.device)
        att = torch.empty((R, w), dtype=bf, device=self.device)
        act = {"gelu": 1, "quick_gelu": 2, "gelu_tanh": 5}[spec.act]

        def rows(norm, base=None, base_rows=0, y=None, y16=None):
            g_, b_ = (self.w[norm + ".g"], self.w[norm + ".b"]) if norm else (None, None)
            L.check(lib.ovo_row_epilogue(L.ptr(x), R, w, L.ptr(base), base_rows, L.ptr(g_), L.ptr(b_), spec.ln_eps, None, 0, L.ptr(y), L.ptr(y16), None,
                                         L.stream()))
        rows(None, base=self.pos, base_rows=t, y=x)                                    # + positional embedding (rows repeat per text)
        for i in range(self.layers):
            p = f"transformer.resblocks.{i}."
            rows(p + "ln_1", y16=h16)
            qkv = self._gemm(h16, p + "qkv", bf)                                       # [R, 3w] = (q | k | v), heads contiguous inside each
            a = L.Attention()
            a.q, a.k, a.v, a.o = qkv.data_ptr(), qkv.data_ptr() + 2 * w, qkv.data_ptr() + 4 * w, att.data_ptr()
            a.q_sb = a.k_sb = a.v_sb = t * 3 * w
            a.q_sh = a.k_sh = a.v_sh = hd
            a.q_st = a.k_st = a.v_st = 3 * w
            a.o_sb, a.o_sh, a.o_st = t * w, hd, w
            a.B, a.H, a.Tq, a.Tk, a.hd, a.scale, a.causal = b, H, t, t, hd, (0.0 if self.q_prescaled else hd ** -0.5), int(spec.causal)
            L.check(lib.ovo_attention(C.byref(a), L.stream()))
            self._gemm(att, p + "out", f32, add=x, out=x)
            rows(p + "ln_2", y16=h16)
            hidden = self._gemm(h16, p + "fc1", bf, act=act)
            self._gemm(hidden, p + "fc2", f32, add=x, out=x)
        final = torch.empty((R, w), dtype=bf, device=self.device)
        rows("ln_final", y16=final)
        if spec.pool == "last":
            eot = [r * t + t - 1 for r in range(b)]
        else:
            eot = (torch.arange(b, device=self.device) * t + tokens.argmax(dim=-1)).tolist()   # end-of-text = highest id (open_clip)
        pooled = L.gather_rows(final, eot)
        return self._gemm(pooled, "proj", f32, bias="proj.b" in self.w)

    def __call__(self, texts: List[str]) -> torch.Tensor:
        """The `text_encoder` callable `CLIPGenerator` expects: list of strings -> [n, out_dim]."""
        if self.tokenizer is None:
            raise L.OvoHipError("no tokenizer: the BPE vocabulary of open_clip / perception_models is not available offline; "
                                "inject tokenizer=callable(list[str]) -> token ids, or call encode_tokens()")
        return self.encode_tokens(self.tokenizer(list(texts)))
