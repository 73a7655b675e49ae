"""SAM2 prompt encoder (point prompts) + mask decoder on MI355X.

The reference reaches this through the un-vendored `sam2` package: `SAM2AutomaticMaskGenerator.generate` prompts the
decoder with a regular grid of foreground clicks (segment_utils.py:291-308, mask_generator.py:113).  Here every prompt of
the grid is decoded in ONE batch: the token side ([P*8, 256]) and the image side ([P*4096, 256]) of the two-way
transformer are plain row-major matrices, all products run on `ovo_gemm` / `ovo_attention`, and the passes between them
(residual, LayerNorm, "+ positional code", casts, the two transposed convolutions, the hyper-network product) are the
three kernels of csrc/samdec.hip.  State-dict names follow the sam2 repository, like `oracle/sam2_decoder.py`.

What is shared between prompts is computed once: the keys of the first layer (identical for every prompt until the first
image->token update) are projected for a single prompt and broadcast through a zero batch stride; the point tokens of a
fixed prompt grid and the dense positional code are constants of the generator, built at set-up.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from .. import _lib as L

import os
# The per-prompt keys [P * S, C] between the two-way layers are kept in bf16 only: the tensor every product reads is bf16 anyway, and reading it
# back as the residual instead of a second f32 copy saves 1.6 GB of traffic per layer at 256 prompts.  The reference runs the decoder under bf16
# autocast (mask_generator.py:46,112: every nn.Linear output is bf16 there); measured against the fp32 oracle the mask-logit error is unchanged
# (max err / rms 4.1e-2 -> 4.0e-2, tests/test_gpu_sam_decoder.py).  OVO_SAM_RES16=0 keeps the f32 residual stream.
RES16 = os.environ.get("OVO_SAM_RES16", "1") == "1"
PE = "sam_prompt_encoder."
MD = "sam_mask_decoder."


@dataclass(frozen=True)
class SamDecoderSpec:
    name: str
    hidden: int = 256
    heads: int = 8
    mlp_dim: int = 2048
    depth: int = 2
    n_mask_tokens: int = 4
    embed_size: int = 64            # side of the image embedding grid
    image_size: int = 1024          # side of the model input the prompts are expressed in
    iou_hidden: int = 256


SPECS: Dict[str, SamDecoderSpec] = {
    "sam2": SamDecoderSpec("sam2"),
    "sam2_small": SamDecoderSpec("sam2_small", mlp_dim=512, embed_size=16, image_size=256),      # real widths, 256^2 input
    "sam2_test": SamDecoderSpec("sam2_test", hidden=128, heads=4, mlp_dim=256, embed_size=16, image_size=256, iou_hidden=128),
}


def random_state(spec: SamDecoderSpec, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded random weights with the sam2 repository's parameter names (there are no checkpoints offline)."""
    g = torch.Generator().manual_seed(seed)
    c, ci = spec.hidden, spec.hidden // 2
    sd: Dict[str, torch.Tensor] = {}

    def lin(name, out, inp):
        sd[name + ".weight"] = torch.randn(out, inp, generator=g) * inp ** -0.5
        sd[name + ".bias"] = torch.randn(out, generator=g) * 0.05

    def norm(name, n):
        sd[name + ".weight"] = 1.0 + 0.1 * torch.randn(n, generator=g)
        sd[name + ".bias"] = 0.1 * torch.randn(n, generator=g)

    def attn(name, inner):
        for p, (o, i) in (("q_proj", (inner, c)), ("k_proj", (inner, c)), ("v_proj", (inner, c)), ("out_proj", (c, inner))):
            lin(f"{name}.{p}", o, i)

    sd[PE + "pe_layer.positional_encoding_gaussian_matrix"] = torch.randn(2, c // 2, generator=g)
    for i in range(4):
        sd[PE + f"point_embeddings.{i}.weight"] = torch.randn(1, c, generator=g) * 0.5
    sd[PE + "not_a_point_embed.weight"] = torch.randn(1, c, generator=g) * 0.5
    sd[PE + "no_mask_embed.weight"] = torch.randn(1, c, generator=g) * 0.5
    sd["no_mem_embed"] = torch.randn(1, 1, c, generator=g) * 0.1
    for n in ("obj_score_token", "iou_token"):
        sd[MD + n + ".weight"] = torch.randn(1, c, generator=g) * 0.5
    sd[MD + "mask_tokens.weight"] = torch.randn(spec.n_mask_tokens, c, generator=g) * 0.5
    t = MD + "transformer."
    for i in range(spec.depth):
        b = t + f"layers.{i}."
        attn(b + "self_attn", c)
        attn(b + "cross_attn_token_to_image", ci)
        attn(b + "cross_attn_image_to_token", ci)
        lin(b + "mlp.layers.0", spec.mlp_dim, c)
        lin(b + "mlp.layers.1", c, spec.mlp_dim)
        for k in range(1, 5):
            norm(b + f"norm{k}", c)
    attn(t + "final_attn_token_to_image", ci)
    norm(t + "norm_final_attn", c)
    sd[MD + "output_upscaling.0.weight"] = torch.randn(c, c // 4, 2, 2, generator=g) * c ** -0.5
    sd[MD + "output_upscaling.0.bias"] = torch.randn(c // 4, generator=g) * 0.05
    norm(MD + "output_upscaling.1", c // 4)
    sd[MD + "output_upscaling.3.weight"] = torch.randn(c // 4, c // 8, 2, 2, generator=g) * (c // 4) ** -0.5
    sd[MD + "output_upscaling.3.bias"] = torch.randn(c // 8, generator=g) * 0.05
    for i in range(spec.n_mask_tokens):
        m = MD + f"output_hypernetworks_mlps.{i}."
        lin(m + "layers.0", c, c); lin(m + "layers.1", c, c); lin(m + "layers.2", c // 8, c)
    lin(MD + "iou_prediction_head.layers.0", spec.iou_hidden, c)
    lin(MD + "iou_prediction_head.layers.1", spec.iou_hidden, spec.iou_hidden)
    lin(MD + "iou_prediction_head.layers.2", spec.n_mask_tokens, spec.iou_hidden)
    lin(MD + "pred_obj_score_head.layers.0", c, c); lin(MD + "pred_obj_score_head.layers.1", c, c); lin(MD + "pred_obj_score_head.layers.2", 1, c)
    return sd


def fourier_pe(coords01: torch.Tensor, gauss: torch.Tensor) -> torch.Tensor:
    """[..., 2] (x, y) in [0, 1] -> [sin | cos](2 pi (2c - 1) G): SAM's random-Fourier positional code (set-up time only)."""
    c = 2.0 * math.pi * ((2.0 * coords01 - 1.0) @ gauss)
    return torch.cat([torch.sin(c), torch.cos(c)], dim=-1)


def point_grid(n_per_side: int) -> torch.Tensor:
    """The regular prompt grid of the automatic mask generator, in [0, 1]^2 (x, y), row-major -> [n*n, 2]."""
    off = 1.0 / (2 * n_per_side)
    t = torch.linspace(off, 1.0 - off, n_per_side, dtype=torch.float64)
    yy, xx = torch.meshgrid(t, t, indexing="ij")
    return torch.stack([xx, yy], dim=-1).reshape(-1, 2)


class HipSamDecoder:
    def __init__(self, spec: SamDecoderSpec, state: Optional[Dict[str, torch.Tensor]] = None, device="cuda", seed: int = 0):
        self.spec, self.device = spec, torch.device(device)
        sd = state if state is not None else random_state(spec, seed)
        self.sd = sd
        c = spec.hidden
        self.w: Dict[str, torch.Tensor] = {}
        self.q_prescaled = L.q_prescale_enabled()

        def up(name, t, dtype):
            self.w[name] = t.to(self.device, dtype).contiguous()

        def lin(name):
            up(name + ".w", sd[name + ".weight"], torch.bfloat16)
            up(name + ".b", sd[name + ".bias"].float(), torch.float32)

        t = MD + "transformer."
        blocks = [t + f"layers.{i}." for i in range(spec.depth)]
        for b in blocks:
            sa = b + "self_attn."
            qkw, qkb = torch.cat([sd[sa + "q_proj.weight"], sd[sa + "k_proj.weight"]]), torch.cat([sd[sa + "q_proj.bias"], sd[sa + "k_proj.bias"]])
            if self.q_prescaled:                                  # token self-attention (ovo_attention): log2 e / sqrt(hd) into the q rows, in f32
                qkw, qkb = L.fold_q_scale(qkw, qkb, c, c // spec.heads)
            up(sa + "qk.w", qkw, torch.bfloat16)
            up(sa + "qk.b", qkb.float(), torch.float32)
            lin(sa + "v_proj"); lin(sa + "out_proj")
            for a in ("cross_attn_token_to_image.", "cross_attn_image_to_token."):
                for p in ("q_proj", "k_proj", "v_proj", "out_proj"):
                    lin(b + a + p)
            lin(b + "mlp.layers.0"); lin(b + "mlp.layers.1")
            for k in range(1, 5):
                up(b + f"norm{k}.g", sd[b + f"norm{k}.weight"].float(), torch.float32)
                up(b + f"norm{k}.b", sd[b + f"norm{k}.bias"].float(), torch.float32)
        for p in ("q_proj", "k_proj", "v_proj", "out_proj"):
            lin(t + "final_attn_token_to_image." + p)
        up(t + "norm_final_attn.g", sd[t + "norm_final_attn.weight"].float(), torch.float32)
        up(t + "norm_final_attn.b", sd[t + "norm_final_attn.bias"].float(), torch.float32)
        for i, key in ((0, "up1"), (3, "up2")):                   # ConvTranspose2d [Cin, Cout, 2, 2] -> GEMM weight [(dy, dx, co), Cin]
            wt = sd[MD + f"output_upscaling.{i}.weight"]
            up(key + ".w", wt.permute(2, 3, 1, 0).reshape(4 * wt.shape[1], wt.shape[0]), torch.bfloat16)
            up(key + ".b", sd[MD + f"output_upscaling.{i}.bias"].float(), torch.float32)
        up("up_ln.g", sd[MD + "output_upscaling.1.weight"].float(), torch.float32)
        up("up_ln.b", sd[MD + "output_upscaling.1.bias"].float(), torch.float32)
        for i in range(spec.n_mask_tokens):
            for j in range(3):
                lin(MD + f"output_hypernetworks_mlps.{i}.layers.{j}")
        for j in range(3):
            lin(MD + f"iou_prediction_head.layers.{j}")
        # constants of the generator
        gauss = sd[PE + "pe_layer.positional_encoding_gaussian_matrix"].double()
        s = spec.embed_size
        tt = (torch.arange(s, dtype=torch.float64) + 0.5) / s
        yy, xx = torch.meshgrid(tt, tt, indexing="ij")
        up("key_pe", fourier_pe(torch.stack([xx, yy], dim=-1), gauss).reshape(s * s, c).float(), torch.float32)
        dense = sd[PE + "no_mask_embed.weight"].reshape(1, c).float()
        if "no_mem_embed" in sd:                                  # SAM2ImagePredictor adds it to the lowest-resolution feature
            dense = dense + sd["no_mem_embed"].reshape(1, c).float()
        up("dense", dense, torch.float32)
        # Image-side projections once the prompts have diverged (layers >= 1 and the final attention): K | V (| Q of the image -> token
        # attention) read the same keys, so they are ONE product over them; the positional code enters as its own projection, a per-pixel
        # constant added in the epilogue (ovo_gemm_periodic): k_proj(keys + pe) = keys . Wk^T + (pe . Wk^T)[pixel].
        pe64 = self.w["key_pe"].double().cpu()

        def fuse(name, parts):                                     # parts: (projection, adds the positional code?)
            ws = [sd[pn + ".weight"] for pn, _ in parts]
            up(name + ".w", torch.cat(ws), torch.bfloat16)
            up(name + ".b", torch.cat([sd[pn + ".bias"] for pn, _ in parts]).float(), torch.float32)
            cols = [pe64 @ w.to(torch.bfloat16).double().T if with_pe else torch.zeros(pe64.shape[0], w.shape[0], dtype=torch.float64)
                    for w, (_, with_pe) in zip(ws, parts)]
            up(name + ".add", torch.cat(cols, dim=1).float(), torch.float32)

        for i in range(1, spec.depth):
            b = blocks[i]
            fuse(b + "kvq", [(b + "cross_attn_token_to_image.k_proj", True), (b + "cross_attn_token_to_image.v_proj", False),
                             (b + "cross_attn_image_to_token.q_proj", True)])
        fuse(t + "final_kv", [(t + "final_attn_token_to_image.k_proj", True), (t + "final_attn_token_to_image.v_proj", False)])
        self._gauss = gauss
        self.tokens0: Optional[torch.Tensor] = None
        self.tok16: Optional[torch.Tensor] = None
        self.P = 0

    # ------------------------------------------------------------------ prompts (set-up)
    def set_points(self, points_xy: torch.Tensor, labels: Optional[torch.Tensor] = None) -> None:
        """points [P, 2] (x, y) in pixels of the image_size^2 model input, one click per prompt; labels [P] (default 1).
        Builds the constant token matrix [P, 8, C] = (obj, iou, 4 mask tokens, click, "not a point" padding)."""
        sd, spec = self.sd, self.spec
        pts = points_xy.double().cpu()
        p = pts.shape[0]
        lab = torch.ones(p, dtype=torch.long) if labels is None else labels.long().cpu()
        pe = fourier_pe((pts + 0.5) / float(spec.image_size), self._gauss).float()
        emb = torch.stack([sd[PE + f"point_embeddings.{i}.weight"][0] for i in (0, 1)]).float()
        click = pe + emb[lab]
        out_tok = torch.cat([sd[MD + "obj_score_token.weight"], sd[MD + "iou_token.weight"], sd[MD + "mask_tokens.weight"]], 0).float()
        pad = sd[PE + "not_a_point_embed.weight"].float()
        tokens = torch.cat([out_tok[None].expand(p, -1, -1), click[:, None], pad[None].expand(p, -1, -1)], dim=1)
        self.tokens0 = tokens.reshape(-1, spec.hidden).to(self.device).contiguous()
        self.tok16 = self.tokens0.to(torch.bfloat16)
        self.P, self.T = p, tokens.shape[1]

    def set_point_grid(self, n_per_side: int) -> torch.Tensor:
        pts = point_grid(n_per_side) * self.spec.image_size
        self.set_points(pts)
        return pts

    # ------------------------------------------------------------------ launches
    def _gemm(self, a: torch.Tensor, wname: str, out_dtype, act: int = 0, add: Optional[torch.Tensor] = None,
              out: Optional[torch.Tensor] = None, rows: Optional[int] = None, lda: Optional[int] = None, ldc: Optional[int] = None,
              a_off: int = 0, c_off: int = 0, bias: bool = True, add_rows: int = 0) -> torch.Tensor:
        w = self.w[wname + ".w"]
        m = a.shape[0] if rows is None else rows
        n, k = w.shape
        if out is None:
            out = torch.empty((m, n), dtype=out_dtype, device=a.device)
        g = L.Gemm()
        g.A, g.lda, g.W, g.ldw = a.data_ptr() + a_off * a.element_size(), (a.stride(0) if lda is None else lda), w.data_ptr(), k
        g.bias = self.w[wname + ".b"].data_ptr() if bias else None
        g.C, g.ldc = out.data_ptr() + c_off * out.element_size(), (out.stride(0) if ldc is None else ldc)
        g.add, g.ld_add = (add.data_ptr(), add.stride(0)) if add is not None else (None, 0)
        g.M, g.N, g.K = m, n, k
        g.in_dtype, g.out_dtype, g.act, g.alpha = 2, L.DTYPE_CODE[out.dtype], act, 1.0
        if add_rows:                                              # add[m % add_rows]: a per-pixel constant shared by every prompt
            L.check(L.load().ovo_gemm_periodic(C.byref(g), add_rows, L.stream()))
        else:
            L.check(L.load().ovo_gemm(C.byref(g), L.stream()))
        return out

    def _image_proj(self, a16: torch.Tensor, wname: str, groups) -> torch.Tensor:
        """bf16 [rows, N] = a16 . W^T + bias + add[row % S] for the fused image-side projection `wname` (see `fuse`), computed as one
        LDS-resident-weight stream per column group (`groups`: widths summing to N; ovo_sam_linear) or, for widths that has no
        instantiation for, as one tiled GEMM with the periodic add."""
        w, bias, add = self.w[wname + ".w"], self.w[wname + ".b"], self.w[wname + ".add"]
        n, k = w.shape
        rows, S = a16.shape[0], add.shape[0]
        out = torch.empty((rows, n), dtype=torch.bfloat16, device=a16.device)
        lib, lo = L.load(), 0
        for width in groups:
            rc = lib.ovo_sam_linear(L.ptr(a16), C.c_void_p(w.data_ptr() + 2 * lo * k), C.c_void_p(bias.data_ptr() + 4 * lo),
                                    C.c_void_p(add.data_ptr() + 4 * lo), S, n, C.c_void_p(out.data_ptr() + 2 * lo), n, rows, width, k, L.stream())
            if rc == L.E_UNSUPPORTED:
                return self._gemm(a16, wname, torch.bfloat16, add=add, add_rows=S, out=out)
            L.check(rc)
            lo += width
        return out

    @staticmethod
    def _attn(q, k, v, o, B, H, Tq, Tk, hd, qs, ks, vs, os_, prescaled=False):
        """q/k/v/o: (tensor, element offset); *s: (batch, head, token) strides in elements.  prescaled: q already carries
        log2(e) / sqrt(hd) (folded into its projection at load time) -> scale 0 = no factor inside the kernel."""
        a = L.Attention()
        a.q, a.k, a.v, a.o = (t.data_ptr() + off * t.element_size() for t, off in (q, k, v, o))
        a.q_sb, a.q_sh, a.q_st = qs
        a.k_sb, a.k_sh, a.k_st = ks
        a.v_sb, a.v_sh, a.v_st = vs
        a.o_sb, a.o_sh, a.o_st = os_
        a.B, a.H, a.Tq, a.Tk, a.hd, a.scale = B, H, Tq, Tk, hd, (0.0 if prescaled else hd ** -0.5)
        L.check(L.load().ovo_attention(C.byref(a), L.stream()))

    def _rows(self, x, R, Cc, *, base=None, base_rows=0, norm=None, eps=1e-5, pe=None, pe_rows=0, y=None, y16=None, ype16=None):
        g = self.w[norm + ".g"] if norm else None
        b = self.w[norm + ".b"] if norm else None
        L.check(L.load().ovo_row_epilogue(L.ptr(x), R, Cc, L.ptr(base), base_rows, L.ptr(g), L.ptr(b), eps, L.ptr(pe), pe_rows,
                                          L.ptr(y), L.ptr(y16), L.ptr(ype16), L.stream()))

    # ------------------------------------------------------------------ forward
    def forward(self, image_embed: torch.Tensor, feat_s1: torch.Tensor, feat_s0: torch.Tensor, multimask: bool = True
                ) -> Tuple[torch.Tensor, torch.Tensor]:
        """image_embed f32 [s, s, C] (NHWC, as HipHiera emits it), feat_s1 f32 [2s, 2s, C/4], feat_s0 f32 [4s, 4s, C/8]
        -> (mask logits f32 [P, 3 | 4, 4s, 4s], predicted IoU f32 [P, 3 | 4]) for the prompts of set_points()."""
        if self.tokens0 is None:
            raise L.OvoHipError("set_points() / set_point_grid() first")
        spec = self.spec
        c, ci, H, s = spec.hidden, spec.hidden // 2, spec.heads, spec.embed_size
        S, P, T = s * s, self.P, self.T
        dev = self.device
        # the per-prompt keys live as bf16 only where the fused out-projection + norm kernel exists (hidden 256 / 128, samfuse.hip); at any
        # other width the unfused chain needs the f32 stream as its residual operand (`force_unfused`: tests)
        res16_on = RES16 and c in (256, 128) and not getattr(self, "force_unfused", False)
        emb = L.dev(image_embed.reshape(S, c), torch.float32, "image_embed")
        f1 = L.dev(feat_s1.reshape(4 * S, c // 4), torch.float32, "feat_s1")
        f0 = L.dev(feat_s0.reshape(16 * S, c // 8), torch.float32, "feat_s0")
        bf, f32 = torch.bfloat16, torch.float32
        lib = L.load()
        key_pe, tok0 = self.w["key_pe"], self.tokens0
        R = P * T

        # keys of layer 0: shared by every prompt
        keys0 = torch.empty((S, c), dtype=f32, device=dev)
        k16 = torch.empty((S, c), dtype=bf, device=dev)
        kpe16 = torch.empty((S, c), dtype=bf, device=dev)
        self._rows(emb, S, c, base=self.w["dense"], base_rows=1, pe=key_pe, pe_rows=S, y=keys0, y16=k16, ype16=kpe16)
        keys = None                                               # f32 [P*S, c] once the prompts diverge
        shared = True

        q = torch.empty((R, c), dtype=f32, device=dev)
        q16 = torch.empty((R, c), dtype=bf, device=dev)
        qpe16 = torch.empty((R, c), dtype=bf, device=dev)
        o_tok = torch.empty((R, c), dtype=bf, device=dev)
        hd_s, hd_c = c // H, ci // H
        t = MD + "transformer."

        def t2i_attention(tq, K, V, kv_sb, kv_st, o):
            rc = lib.ovo_sam_t2i_attention(L.ptr(tq), K, V, kv_sb, kv_st, L.ptr(o), P, S, T, H, hd_c ** -0.5, L.stream()) if hd_c == 16 \
                else L.E_UNSUPPORTED
            return rc

        def token_to_image(pre, qin16, kv=None):
            """kv: the fused per-prompt projection bf16 [P*S, n * ci] (K | V | ...), or None while the keys are shared (layer 0)."""
            nonlocal q
            tq = self._gemm(qin16, pre + "q_proj", bf)                                    # [R, ci]
            if kv is None:
                K = self._gemm(kpe16, pre + "k_proj", bf)                                 # [S, ci]
                V = self._gemm(k16, pre + "v_proj", bf)
                kt_, vt_, kb, ks = (K, 0), (V, 0), 0, ci
            else:
                kt_, vt_, kb, ks = (kv, 0), (kv, ci), S * kv.shape[1], kv.shape[1]
            o = torch.empty((R, ci), dtype=bf, device=dev)
            kptr, vptr = (C.c_void_p(t_.data_ptr() + off * 2) for t_, off in (kt_, vt_))
            rc = t2i_attention(tq, kptr, vptr, kb, ks, o)
            if rc == L.E_UNSUPPORTED:
                self._attn((tq, 0), kt_, vt_, (o, 0), P, H, T, S, hd_c, (T * ci, hd_c, ci), (kb, hd_c, ks), (kb, hd_c, ks), (T * ci, hd_c, ci))
            else:
                L.check(rc)
            self._gemm(o, pre + "out_proj", f32, add=q, out=q)

        for i in range(spec.depth):
            b = t + f"layers.{i}."
            # per-prompt keys: K | V of the token -> image attention and Q of the image -> token attention in one product over them
            kvq = None if shared else self._image_proj(k16, b + "kvq", (2 * ci, ci))                            # [P*S, 3 ci]
            # ---- self attention on the tokens (first layer: no positional code, no residual)
            qk = self._gemm(self.tok16 if i == 0 else qpe16, b + "self_attn.qk", bf)      # [R, 2c]
            v = self._gemm(self.tok16 if i == 0 else q16, b + "self_attn.v_proj", bf)     # [R, c]
            self._attn((qk, 0), (qk, c), (v, 0), (o_tok, 0), P, H, T, T, hd_s, (T * 2 * c, hd_s, 2 * c), (T * 2 * c, hd_s, 2 * c),
                       (T * c, hd_s, c), (T * c, hd_s, c), prescaled=self.q_prescaled)
            self._gemm(o_tok, b + "self_attn.out_proj", f32, add=None if i == 0 else q, out=q)
            self._rows(q, R, c, norm=b + "norm1", pe=tok0, pe_rows=R, y=q, y16=q16, ype16=qpe16)
            # ---- tokens attend to the image
            token_to_image(b + "cross_attn_token_to_image.", qpe16, kvq)
            self._rows(q, R, c, norm=b + "norm2", y=q, y16=q16)
            # ---- MLP
            h = self._gemm(q16, b + "mlp.layers.0", bf, act=3)
            self._gemm(h, b + "mlp.layers.1", f32, add=q, out=q)
            self._rows(q, R, c, norm=b + "norm3", pe=tok0, pe_rows=R, y=q, y16=q16, ype16=qpe16)
            # ---- image attends to the tokens
            a = b + "cross_attn_image_to_token."
            kt = self._gemm(qpe16, a + "k_proj", bf)                                      # [R, ci]
            vt = self._gemm(q16, a + "v_proj", bf)
            if shared:
                qi, q_off, q_sb, q_st = self._gemm(kpe16, a + "q_proj", bf), 0, 0, ci     # [S, ci]
            else:
                qi, q_off, q_sb, q_st = kvq, 2 * ci, S * 3 * ci, 3 * ci
            oi = torch.empty((P * S, ci), dtype=bf, device=dev)
            if hd_c == 16 and T <= 16 and 256 % H == 0:           # S queries x 8 keys: the dedicated HBM-bound kernel
                L.check(lib.ovo_sam_i2t_attention(C.c_void_p(qi.data_ptr() + 2 * q_off), q_sb, q_st, L.ptr(kt), L.ptr(vt), L.ptr(oi), P, S, T, H,
                                                  hd_c ** -0.5, L.stream()))
            else:
                self._attn((qi, q_off), (kt, 0), (vt, 0), (oi, 0), P, H, S, T, hd_c, (q_sb, hd_c, q_st), (T * ci, hd_c, ci),
                           (T * ci, hd_c, ci), (S * ci, hd_c, ci))
            # out-projection + residual + norm4 (+ the bf16 copies the next products read): one fused launch (samfuse.hip); widths it
            # does not cover run the product and the row pass separately
            last = i == spec.depth - 1
            if shared:                                            # the prompts diverge here: materialise per-prompt keys
                keys = None if last or res16_on else torch.empty((P * S, c), dtype=f32, device=dev)
                k16 = torch.empty((P * S, c), dtype=bf, device=dev)
                kpe16 = None                                      # (keys + pe) is never formed per prompt: see `fuse` in __init__
                res, res16, res_rows = keys0, None, S
            elif res16_on:
                res, res16, res_rows, k16 = None, k16, P * S, torch.empty((P * S, c), dtype=bf, device=dev)
            else:
                res, res16, res_rows = keys, None, P * S
            y32 = None if last else keys                          # the f32 residual stream is not read after the last layer
            rc = L.E_UNSUPPORTED if getattr(self, "force_unfused", False) else \
                lib.ovo_sam_proj_ln(L.ptr(oi), L.ptr(self.w[a + "out_proj.w"]), L.ptr(self.w[a + "out_proj.b"]), L.ptr(res), L.ptr(res16), res_rows,
                                    L.ptr(self.w[b + "norm4.g"]), L.ptr(self.w[b + "norm4.b"]), 1e-5, L.ptr(key_pe), S, L.ptr(y32), L.ptr(k16),
                                    L.ptr(kpe16), P * S, c, ci, L.stream())
            if rc == L.E_UNSUPPORTED:                             # product and row pass separately (f32 stream: res16_on is off at these widths)
                if shared:
                    x = self._gemm(oi, a + "out_proj", f32)
                    self._rows(x, P * S, c, base=keys0, base_rows=S, norm=b + "norm4", pe=key_pe, pe_rows=S, y=y32, y16=k16, ype16=kpe16)
                else:
                    self._gemm(oi, a + "out_proj", f32, add=keys, out=keys)
                    self._rows(keys, P * S, c, norm=b + "norm4", pe=key_pe, pe_rows=S, y=None if last else keys, y16=k16, ype16=kpe16)
            else:
                L.check(rc)
            shared = False
        kv = None if shared else self._image_proj(k16, t + "final_kv", (2 * ci,))                               # [P*S, 2 ci]
        token_to_image(t + "final_attn_token_to_image.", qpe16, kv)
        self._rows(q, R, c, norm=t + "norm_final_attn", y=q, y16=q16)

        # ---- heads: IoU prediction and hyper-network rows straight from strided token rows
        nm = spec.n_mask_tokens
        x = self._gemm(q16, MD + "iou_prediction_head.layers.0", bf, act=3, rows=P, lda=T * c, a_off=1 * c)
        x = self._gemm(x, MD + "iou_prediction_head.layers.1", bf, act=3)
        iou = self._gemm(x, MD + "iou_prediction_head.layers.2", f32, act=4)              # [P, nm]
        hyper = torch.empty((P, nm, c // 8), dtype=f32, device=dev)
        for m in range(nm):
            pre = MD + f"output_hypernetworks_mlps.{m}."
            x = self._gemm(q16, pre + "layers.0", bf, act=3, rows=P, lda=T * c, a_off=(2 + m) * c)
            x = self._gemm(x, pre + "layers.1", bf, act=3)
            self._gemm(x, pre + "layers.2", f32, out=hyper, ldc=nm * (c // 8), c_off=m * (c // 8))
        # ---- upscaling: two transposed convolutions as skinny products fused with what follows them (LayerNorm2d + GELU; GELU + the
        # hyper-network product), samfuse.hip
        up1 = torch.empty((P * 4 * S, c // 4), dtype=bf, device=dev)
        rc = lib.ovo_sam_up1_ln(L.ptr(k16), L.ptr(self.w["up1.w"]), L.ptr(self.w["up1.b"]), L.ptr(f1), L.ptr(self.w["up_ln.g"]),
                                L.ptr(self.w["up_ln.b"]), 1e-6, P, s, c // 4, c, L.ptr(up1), L.stream())
        if rc == L.E_UNSUPPORTED:
            g1 = self._gemm(k16, "up1", bf, bias=False)                                   # [P*S, 4 * c/4]
            rc = lib.ovo_sam_upscale_ln(L.ptr(g1), L.ptr(self.w["up1.b"]), L.ptr(f1), L.ptr(self.w["up_ln.g"]), L.ptr(self.w["up_ln.b"]), 1e-6,
                                        P, s, c // 4, L.ptr(up1), L.stream())
        L.check(rc)
        first = 1 if multimask else 0
        n_out = nm - first if multimask else 1
        masks = torch.empty((P, n_out, 4 * s, 4 * s), dtype=f32, device=dev)
        hy = hyper if multimask else hyper[:, :1].contiguous()
        n_m, fst = (nm, 1) if multimask else (1, 0)
        rc = lib.ovo_sam_up2_masks(L.ptr(up1), L.ptr(self.w["up2.w"]), L.ptr(self.w["up2.b"]), L.ptr(f0), L.ptr(hy), n_m, fst, P, 2 * s, c // 8,
                                   c // 4, L.ptr(masks), L.stream())
        if rc == L.E_UNSUPPORTED:
            g2 = self._gemm(up1, "up2", bf, bias=False)                                   # [P*4S, 4 * c/8]
            rc = lib.ovo_sam_upscale_masks(L.ptr(g2), L.ptr(self.w["up2.b"]), L.ptr(f0), L.ptr(hy), n_m, fst, P, 2 * s, c // 8, L.ptr(masks),
                                           L.stream())
        L.check(rc)
        return (masks, iou[:, 1:]) if multimask else (masks, iou[:, :1])
