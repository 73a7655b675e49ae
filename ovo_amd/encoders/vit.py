"""Vision transformer image encoder on MI355X.

Replaces the third-party towers the reference calls -- open_clip `model.encode_image`
(clip_generator.py:112-122) and perception_models `visual.forward_features(x, norm=True)`
(textregion.py:141-142) -- with `ovo_vit_forward` from libovo_hip.so (MFMA GEMMs, fused attention, fp32
residual stream).  Weights use open_clip's VisionTransformer state-dict names, so a real checkpoint's
`visual.*` tensors load unchanged; offline (no hub access) `random_state` gives seeded weights of the same
architecture, which is all throughput and parity-vs-oracle need.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .. import _lib as L

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


@dataclass(frozen=True)
class ViTSpec:
    name: str
    image_size: int
    patch: int
    width: int
    layers: int
    heads: int
    mlp_dim: int
    out_dim: int
    act: str = "gelu"                 # "gelu" | "quick_gelu" (DFN "-qg" cards, clip_utils.py:57-60) | "gelu_tanh" (SigLIP)
    pre_ln: bool = True
    use_rope: bool = False            # perception_models Rope2D
    cls_token: bool = True
    ln_eps: float = 1e-5
    mean: Tuple[float, float, float] = CLIP_MEAN
    std: Tuple[float, float, float] = CLIP_STD
    attn_pool_heads: int = 0          # > 0: PE attention-pool head (only W_v / W_o / proj are used by TextRegion)
    map_pool: bool = False            # SigLIP: timm AttentionPoolLatent head instead of class token + proj
    patch_bias: bool = False          # conv1 has a bias (timm / SigLIP patch embedding)
    # the card's open_clip preprocess (open_clip 2.32 `image_transform`, of which clip_utils.py:83-84 keeps Resize / CenterCrop /
    # Normalize): "shortest" = Resize(size) on the shorter side + CenterCrop(size) (OpenAI / LAION / DFN cards), "squash" =
    # Resize((size, size)) (the timm-hub cards: SigLIP, PE); torchvision 0.20 resizes tensors antialiased
    resize_mode: str = "shortest"
    interpolation: str = "bicubic"
    # Rope2D conventions (perception_models core/vision_encoder/rope.py, not available offline -- see `rope_tables`): the two switches a
    # maintainer with the upstream package flips if `tools/check_rope.py` shows a mismatch
    rope_cls_offset: int = 1          # patch positions start at 1 when there is a class token ("leave space for the cls token to be (0, 0)")
    rope_axis_order: str = "xy"       # first half of a head's channels rotates with the COLUMN (x), second half with the ROW (y)

    @property
    def grid(self) -> int:
        return self.image_size // self.patch

    @property
    def tokens(self) -> int:
        return self.grid * self.grid + int(self.cls_token)

    @property
    def kpad(self) -> int:
        return (3 * self.patch * self.patch + 31) // 32 * 32

    @property
    def mlp_pad(self) -> int:
        """Hidden width the kernels run with: zero rows / columns up to a multiple of 32 (SigLIP so400m has 4304)."""
        return (self.mlp_dim + 31) // 32 * 32

    def flops_per_image(self) -> float:
        """Dense FLOPs of one forward (SURVEY.md §8d formula: 24 N d^2 L + 4 N^2 d L + patch embed)."""
        n, d, l = self.tokens, self.width, self.layers
        r = self.mlp_dim / d
        return (2 * n * d * d * (4 + 2 * r) + 4 * n * n * d) * l + 2 * (self.grid ** 2) * 3 * self.patch ** 2 * d


# model cards the reference can select (clip_utils.py:53-75, ovo.yaml:46)
SPECS: Dict[str, ViTSpec] = {
    "ViT-B-16-qg": ViTSpec("ViT-B-16-qg", 224, 16, 768, 12, 12, 3072, 512, act="quick_gelu"),
    "ViT-L-14-qg": ViTSpec("ViT-L-14-qg", 224, 14, 1024, 24, 16, 4096, 768, act="quick_gelu"),
    "ViT-H-14": ViTSpec("ViT-H-14", 224, 14, 1280, 32, 16, 5120, 1024),
    "ViT-H-14-qg": ViTSpec("ViT-H-14-qg", 224, 14, 1280, 32, 16, 5120, 1024, act="quick_gelu"),
    "PE-Core-L14-336": ViTSpec("PE-Core-L14-336", 336, 14, 1024, 24, 16, 4096, 1024, use_rope=True,
                               mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5), attn_pool_heads=8, resize_mode="squash", interpolation="bilinear"),
    "ViT-H-14-378qg": ViTSpec("ViT-H-14-378qg", 378, 14, 1280, 32, 16, 5120, 1024, act="quick_gelu"),
    # SigLIP so400m towers (open_clip "ViT-SO400M-14-SigLIP[-384]", clip_utils.py:62-75): no class token, no ln_pre,
    # tanh-GELU, LayerNorm eps 1e-6, attention-pool head, no output projection
    "SigLIP": ViTSpec("SigLIP", 224, 14, 1152, 27, 16, 4304, 1152, act="gelu_tanh", pre_ln=False, cls_token=False, ln_eps=1e-6,
                      mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5), map_pool=True, patch_bias=True, resize_mode="squash"),
    "SigLIP-384": ViTSpec("SigLIP-384", 384, 14, 1152, 27, 16, 4304, 1152, act="gelu_tanh", pre_ln=False, cls_token=False,
                          ln_eps=1e-6, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5), map_pool=True, patch_bias=True, resize_mode="squash"),   # 27x27 patches: the
                                                                                     # stride-14 conv ignores the last 6 rows / columns
    "SigLIP2-384": ViTSpec("SigLIP2-384", 384, 16, 1152, 27, 16, 4304, 1152, act="gelu_tanh", pre_ln=False, cls_token=False,
                           ln_eps=1e-6, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5), map_pool=True, patch_bias=True, resize_mode="squash"),
    # reduced shapes for tests
    "tiny-siglip": ViTSpec("tiny-siglip", 56, 14, 128, 2, 4, 432, 128, act="gelu_tanh", pre_ln=False, cls_token=False, ln_eps=1e-6,
                           mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5), map_pool=True, patch_bias=True, resize_mode="squash"),
    "tiny-clip": ViTSpec("tiny-clip", 64, 16, 128, 2, 4, 512, 64, act="quick_gelu"),
    "tiny-pe": ViTSpec("tiny-pe", 84, 14, 128, 2, 4, 512, 128, use_rope=True, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5),
                       attn_pool_heads=4, resize_mode="squash", interpolation="bilinear"),
}


def random_state(spec: ViTSpec, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded fp32 CPU weights with open_clip VisionTransformer names (+ `attn_pool.*` for PE)."""
    g = torch.Generator().manual_seed(seed)
    d, t = spec.width, spec.tokens

    def rn(*shape, std=0.02):
        return torch.randn(*shape, generator=g) * std
    sd = {"conv1.weight": rn(d, 3, spec.patch, spec.patch, std=0.05), "positional_embedding": rn(t, d),
          "ln_post.weight": 1 + rn(d, std=0.05), "ln_post.bias": rn(d, std=0.05)}
    if not spec.map_pool:
        sd["proj"] = rn(d, spec.out_dim, std=d ** -0.5)
    if spec.patch_bias:
        sd["conv1.bias"] = rn(d)
    if spec.cls_token:
        sd["class_embedding"] = rn(d)
    if spec.pre_ln:
        sd["ln_pre.weight"], sd["ln_pre.bias"] = 1 + rn(d, std=0.05), rn(d, std=0.05)
    for i in range(spec.layers):
        p = f"transformer.resblocks.{i}."
        sd[p + "ln_1.weight"], sd[p + "ln_1.bias"] = 1 + rn(d, std=0.05), rn(d, std=0.05)
        sd[p + "ln_2.weight"], sd[p + "ln_2.bias"] = 1 + rn(d, std=0.05), rn(d, std=0.05)
        sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"] = rn(3 * d, d, std=d ** -0.5), rn(3 * d)
        sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"] = rn(d, d, std=d ** -0.5), rn(d)
        sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"] = rn(spec.mlp_dim, d, std=d ** -0.5), rn(spec.mlp_dim)
        sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"] = rn(d, spec.mlp_dim, std=spec.mlp_dim ** -0.5), rn(d)
    if spec.attn_pool_heads:
        sd["attn_pool.probe"] = rn(1, 1, d)
        sd["attn_pool.attn.in_proj_weight"], sd["attn_pool.attn.in_proj_bias"] = rn(3 * d, d, std=d ** -0.5), rn(3 * d)
        sd["attn_pool.attn.out_proj.weight"], sd["attn_pool.attn.out_proj.bias"] = rn(d, d, std=d ** -0.5), rn(d)
        sd["attn_pool.layernorm.weight"], sd["attn_pool.layernorm.bias"] = 1 + rn(d, std=0.05), rn(d, std=0.05)
    if spec.map_pool:
        m = spec.mlp_dim
        sd["attn_pool.latent"] = rn(1, 1, d, std=d ** -0.5)
        sd["attn_pool.q.weight"], sd["attn_pool.q.bias"] = rn(d, d, std=d ** -0.5), rn(d)
        sd["attn_pool.kv.weight"], sd["attn_pool.kv.bias"] = rn(2 * d, d, std=d ** -0.5), rn(2 * d)
        sd["attn_pool.proj.weight"], sd["attn_pool.proj.bias"] = rn(d, d, std=d ** -0.5), rn(d)
        sd["attn_pool.norm.weight"], sd["attn_pool.norm.bias"] = 1 + rn(d, std=0.05), rn(d, std=0.05)
        sd["attn_pool.mlp.fc1.weight"], sd["attn_pool.mlp.fc1.bias"] = rn(m, d, std=d ** -0.5), rn(m)
        sd["attn_pool.mlp.fc2.weight"], sd["attn_pool.mlp.fc2.bias"] = rn(d, m, std=m ** -0.5), rn(d)
    return sd


def rope_tables(spec: ViTSpec, theta: float = 10000.0, cls_offset: Optional[int] = None, axis_order: Optional[str] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """cos / sin f32 [T, head_dim] of PE's 2-D axial rotary embedding in interleaved-pair form.

    perception_models is not vendored in the reference (.gitmodules) and not available offline, so this restates
    `core/vision_encoder/rope.py::Rope2D.update_grid` from its published source as best recalled [upstream-knowledge, UNPINNED]:
      * one 1-D rotary table of head_dim / 2 channels per axis: frequency i = theta^(-i / (head_dim / 4)), i = 0 .. head_dim/4 - 1, each
        shared by the adjacent channel pair (2i, 2i+1), which rotates as (x0, x1) -> (x0 cos - x1 sin, x1 cos + x0 sin);
      * with a class token the grid coordinates are `arange(G) + 1` ("+1 to leave space for the cls token to be (0, 0)")  -> cls_offset 1;
      * `freq = cat([freqs_x, freqs_y], -1)`: the first half of a head rotates with the column, the second with the row       -> axis_order "xy";
      * the class token's row is all zeros (identity rotation).
    Round 1-2 of this build used cls_offset 0 / axis_order "yx"; both conventions are parameters (ViTSpec.rope_cls_offset /
    rope_axis_order), all four combinations are tested HIP-vs-oracle, and `tools/check_rope.py` writes all four for a maintainer to diff
    against the real package."""
    off = spec.rope_cls_offset if cls_offset is None else int(cls_offset)
    order = spec.rope_axis_order if axis_order is None else axis_order
    if order not in ("xy", "yx"):
        raise ValueError("axis_order must be 'xy' or 'yx'")
    hd = spec.width // spec.heads
    quarter = hd // 4
    freqs = theta ** (-torch.arange(quarter, dtype=torch.float32) / quarter)
    pos = torch.arange(spec.grid, dtype=torch.float32) + float(off if spec.cls_token else 0)
    ang = pos[:, None] * freqs[None, :]                                   # [G, hd/4]
    ang = ang.repeat_interleave(2, dim=1)                                 # pairs share the angle -> [G, hd/2]
    ay = ang[:, None, :].expand(spec.grid, spec.grid, hd // 2)            # varies with the row
    ax = ang[None, :, :].expand(spec.grid, spec.grid, hd // 2)            # varies with the column
    full = torch.cat([ax, ay] if order == "xy" else [ay, ax], dim=-1).reshape(spec.grid * spec.grid, hd)
    if spec.cls_token:
        full = torch.cat([torch.zeros(1, hd), full], dim=0)
    return full.cos().contiguous(), full.sin().contiguous()


class HipViT:
    """A ViT whose forward pass is one call into libovo_hip.so."""

    def __init__(self, spec: ViTSpec, state: Optional[Dict[str, torch.Tensor]] = None, device="cuda", seed: int = 0):
        self.spec, self.device = spec, torch.device(device)
        sd = state if state is not None else random_state(spec, seed)
        self._keep: List[torch.Tensor] = []                # device tensors referenced by raw pointers
        d = spec.width

        def mat(t):                                       # bf16 matrix [out, in]
            x = t.detach().to(device=self.device, dtype=torch.bfloat16).contiguous()
            self._keep.append(x)
            return L.ptr(x)

        def vec(t):
            x = t.detach().to(device=self.device, dtype=torch.float32).contiguous()
            self._keep.append(x)
            return L.ptr(x)

        self.q_prescaled = L.q_prescale_enabled()
        # LayerNorm fold (ovo_vit_layer_t.qkv_wf ...): a second, gamma-scaled copy of the QKV / FC1 matrices (+ 340 MB for ViT-L); used by the library for
        # batched forwards only.  OVO_VIT_LNFOLD=0: not prepared (and the library keeps its LayerNorm kernels)
        self.ln_fold = os.environ.get("OVO_VIT_LNFOLD", "1") != "0" and spec.act in ("gelu",) and spec.width % 64 == 0 and spec.width <= 1024
        extra = spec.mlp_pad - spec.mlp_dim               # zero hidden units: act(0) = 0 and their fc2 columns are 0

        def pad_rows(t):
            return t if not extra else torch.cat([t.float(), t.new_zeros((extra,) + tuple(t.shape[1:]), dtype=torch.float32)])

        def pad_cols(t):
            return t if not extra else torch.cat([t.float(), t.new_zeros((t.shape[0], extra), dtype=torch.float32)], dim=1)

        conv = sd["conv1.weight"].reshape(d, -1).float()
        pw = torch.zeros(d, spec.kpad)
        pw[:, :conv.shape[1]] = conv
        self._layers = (L.VitLayer * spec.layers)()
        for i in range(spec.layers):
            p = f"transformer.resblocks.{i}."
            ls1 = sd.get(p + "ls_1.gamma")                 # LayerScale (if any) folds into the projection
            ls2 = sd.get(p + "ls_2.gamma")
            ow, ob = sd[p + "attn.out_proj.weight"].float(), sd[p + "attn.out_proj.bias"].float()
            fw, fb = sd[p + "mlp.c_proj.weight"].float(), sd[p + "mlp.c_proj.bias"].float()
            if ls1 is not None:
                ow, ob = ow * ls1.float()[:, None], ob * ls1.float()
            if ls2 is not None:
                fw, fb = fw * ls2.float()[:, None], fb * ls2.float()
            ly = self._layers[i]
            ly.ln1_g, ly.ln1_b = vec(sd[p + "ln_1.weight"]), vec(sd[p + "ln_1.bias"])
            qw, qb = sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"]
            if self.q_prescaled:                           # log2 e / sqrt(hd) into the q rows, in f32, before the bf16 rounding
                qw, qb = L.fold_q_scale(qw, qb, d, d // spec.heads)
            ly.qkv_w, ly.qkv_b = mat(qw), vec(qb)
            ly.out_w, ly.out_b = mat(ow), vec(ob)
            ly.ln2_g, ly.ln2_b = vec(sd[p + "ln_2.weight"]), vec(sd[p + "ln_2.bias"])
            ly.fc1_w, ly.fc1_b = mat(pad_rows(sd[p + "mlp.c_fc.weight"])), vec(pad_rows(sd[p + "mlp.c_fc.bias"]))
            if self.ln_fold:                               # batched forwards: LayerNorm 1 / 2 live in the QKV / FC1 products (vit.hip)
                wf, bf, cs = L.fold_layernorm(qw, qb, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"])
                ly.qkv_wf, ly.qkv_bf, ly.qkv_cs = mat(wf), vec(bf), vec(cs)
                wf, bf, cs = L.fold_layernorm(pad_rows(sd[p + "mlp.c_fc.weight"]), pad_rows(sd[p + "mlp.c_fc.bias"]), sd[p + "ln_2.weight"], sd[p + "ln_2.bias"])
                ly.fc1_wf, ly.fc1_bf, ly.fc1_cs = mat(wf), vec(bf), vec(cs)
            ly.fc2_w, ly.fc2_b = mat(pad_cols(fw)), vec(fb)
        w = L.VitWeights()
        w.patch_w = mat(pw)
        w.patch_b = vec(sd["conv1.bias"]) if "conv1.bias" in sd else None
        w.prefix = vec(sd["class_embedding"].reshape(1, d)) if spec.cls_token else None
        w.pos = vec(sd["positional_embedding"]) if "positional_embedding" in sd else None
        if spec.pre_ln:
            w.ln_pre_g, w.ln_pre_b = vec(sd["ln_pre.weight"]), vec(sd["ln_pre.bias"])
        w.ln_post_g, w.ln_post_b = vec(sd["ln_post.weight"]), vec(sd["ln_post.bias"])
        if "proj" in sd:
            w.proj_w = mat(sd["proj"].float().t())
        if spec.map_pool:
            a = "attn_pool."
            q = torch.nn.functional.linear(sd[a + "latent"].float().reshape(1, d), sd[a + "q.weight"].float(), sd[a + "q.bias"].float())
            if self.q_prescaled:
                q = q * (L.LOG2E / float(d // spec.heads) ** 0.5)
            w.map_q = mat(q.reshape(d))
            w.map_kv_w, w.map_kv_b = mat(sd[a + "kv.weight"]), vec(sd[a + "kv.bias"])
            w.map_proj_w, w.map_proj_b = mat(sd[a + "proj.weight"]), vec(sd[a + "proj.bias"])
            w.map_ln_g, w.map_ln_b = vec(sd[a + "norm.weight"]), vec(sd[a + "norm.bias"])
            w.map_fc1_w, w.map_fc1_b = mat(pad_rows(sd[a + "mlp.fc1.weight"])), vec(pad_rows(sd[a + "mlp.fc1.bias"]))
            w.map_fc2_w, w.map_fc2_b = mat(pad_cols(sd[a + "mlp.fc2.weight"].float())), vec(sd[a + "mlp.fc2.bias"])
        if spec.use_rope:
            cos, sin = rope_tables(spec)
            w.rope_cos, w.rope_sin = vec(cos), vec(sin)
        w.layers = C.cast(self._layers, C.POINTER(L.VitLayer))
        self._weights = w
        self._cfg = {}
        self._ws: Optional[torch.Tensor] = None
        self.split_streams = bool(os.environ.get("OVO_VIT_SPLIT"))     # two half-batch chains on two streams (_forward_split)
        self._side = None
        self._ws2 = None
        self.proj = sd["proj"].detach().to(self.device, torch.float32) if "proj" in sd else None
        self.pool_weights = None
        if spec.attn_pool_heads:
            self.pool_weights = {k[len("attn_pool."):]: v.detach().float() for k, v in sd.items() if k.startswith("attn_pool.")}

    def _config(self, pool: int) -> L.VitConfig:
        c = self._cfg.get(pool)
        if c is None:
            s = self.spec
            c = L.VitConfig(s.image_size, s.patch, s.width, s.layers, s.heads, s.mlp_pad, s.out_dim, int(s.cls_token),
                            {"gelu": 1, "quick_gelu": 2, "gelu_tanh": 5}[s.act], int(s.pre_ln), int(s.use_rope), pool, s.kpad, s.ln_eps, int(self.q_prescaled))
            self._cfg[pool] = c
        return c

    def _workspace(self, cfg, batch: int):
        need = L.load().ovo_vit_workspace_bytes(C.byref(cfg), batch)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws, need

    # ---------------------------------------------------------------- preprocessing
    def preprocess(self, image: torch.Tensor, crops: Optional[Sequence[Tuple[int, int, int, int]]] = None,
                   scale: float = 1.0, antialias: bool = True, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """CHW image (u8 0..255 or f32), or an HWC u8 frame (last dim 3: read in place, no permute copy) -> f32 [len(crops), 3, S, S]:
        squash-resize each crop (y0, x0, h, w) to the
        model resolution (bilinear, antialiased like torchvision's tensor Resize) and normalise with mean / std.
        `scale` multiplies pixel values first (1/255 for u8 input expected in [0, 1])."""
        s = self.spec
        img = L.dev(image, image.dtype, "image")
        if img.dtype not in (torch.uint8, torch.float32):
            raise L.OvoHipError("image must be u8 or f32")
        hwc = img.dtype == torch.uint8 and img.shape[-1] == 3 and img.shape[0] != 3
        h, w = (img.shape[0], img.shape[1]) if hwc else (img.shape[1], img.shape[2])
        crops = list(crops) if crops is not None else [(0, 0, h, w)]
        if out is None:
            out = torch.empty((len(crops), 3, s.image_size, s.image_size), dtype=torch.float32, device=img.device)
        mean, std = (C.c_float * 3)(*s.mean), (C.c_float * 3)(*s.std)
        lib = L.load()
        code = 4 if hwc else L.DTYPE_CODE[img.dtype]
        for i, (y0, x0, ch, cw) in enumerate(crops):
            L.check(lib.ovo_resize_normalize(L.ptr(img), code, 3, h, w, y0, x0, ch, cw, L.ptr(out[i]),
                                             s.image_size, s.image_size, int(antialias), float(scale), mean, std, L.stream()))
        return out

    def preprocess_batch(self, images: Sequence[torch.Tensor], crops: Optional[Sequence[Tuple[int, int, int, int]]] = None, scale: float = 1.0,
                         antialias: bool = True, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`preprocess` of SEVERAL frames (MI355X extension: the look-ahead encoders take a group of keyframes): out[i * len(crops) + k] = crop k of
        image i.  Frames of one size / layout go down as ONE launch (`ovo_resize_normalize_batch`); anything else frame by frame.  Same arithmetic."""
        s = self.spec
        imgs = [L.dev(im, im.dtype, "image") for im in images]
        first = imgs[0]
        same = all(im.dtype == first.dtype and im.shape == first.shape and im.device == first.device for im in imgs) and first.dtype in (torch.uint8, torch.float32)
        hwc = first.dtype == torch.uint8 and first.shape[-1] == 3 and first.shape[0] != 3
        h, w = (first.shape[0], first.shape[1]) if hwc else (first.shape[1], first.shape[2])
        given = list(crops) if crops is not None else None          # (None stays None in the frame-by-frame path: every frame's OWN extent)
        crops = given if given is not None else [(0, 0, h, w)]
        nc = len(crops)
        if out is None:
            out = torch.empty((len(imgs) * nc, 3, s.image_size, s.image_size), dtype=torch.float32, device=first.device)
        if not same:
            for i, im in enumerate(imgs):
                self.preprocess(im, given, scale, antialias, out=out[i * nc:(i + 1) * nc])
            return out
        mean, std = (C.c_float * 3)(*s.mean), (C.c_float * 3)(*s.std)
        srcs = (C.c_void_p * len(imgs))(*[im.data_ptr() for im in imgs])
        rects = (C.c_int32 * (4 * nc))(*[int(v) for r in crops for v in r])
        L.check(L.load().ovo_resize_normalize_batch(srcs, len(imgs), 4 if hwc else L.DTYPE_CODE[first.dtype], 3, h, w, rects, nc, L.ptr(out), s.image_size,
                                                    s.image_size, int(antialias), float(scale), mean, std, L.stream()))
        return out

    def clip_window(self, h: int, w: int) -> Tuple[int, int, int, int]:
        """(virt_h, virt_w, top, left) of the card's open_clip transform on an h x w image: torchvision `Resize(S)` puts the shorter side
        at S and the longer at int(S * long / short); `CenterCrop(S)` starts at int(round((size - S) / 2)).  "squash": (S, S, 0, 0)."""
        S = self.spec.image_size
        if self.spec.resize_mode == "squash":
            return S, S, 0, 0
        vh, vw = (int(S * h / w), S) if w <= h else (S, int(S * w / h))
        return vh, vw, int(round((vh - S) / 2.0)), int(round((vw - S) / 2.0))

    def preprocess_clip(self, images: torch.Tensor, scale: float = 1.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The reference's `self.preprocess` (clip_generator.py:119, clip_utils.py:83-84) for a batch [B, 3, h, w] (u8 or f32): the card's
        Resize (antialiased bicubic / bilinear, shorter side or squash) + CenterCrop + Normalize, one launch per image, writing only the
        kept window."""
        s = self.spec
        x = L.dev(images, images.dtype, "images")
        if x.dtype not in (torch.uint8, torch.float32):
            raise L.OvoHipError("images must be u8 or f32")
        b, _, h, w = x.shape
        if out is None:
            out = torch.empty((b, 3, s.image_size, s.image_size), dtype=torch.float32, device=x.device)
        vh, vw, top, left = self.clip_window(h, w)
        filt = 2 if s.interpolation == "bicubic" else 1
        mean, std = (C.c_float * 3)(*s.mean), (C.c_float * 3)(*s.std)
        lib = L.load()
        for i in range(b):
            L.check(lib.ovo_resize_window_normalize(L.ptr(x[i]), L.DTYPE_CODE[x.dtype], 3, h, w, 0, 0, h, w, L.ptr(out[i]), s.image_size, s.image_size,
                                                    vh, vw, top, left, filt, float(scale), mean, std, L.stream()))
        return out

    # ---------------------------------------------------------------- forward
    def forward(self, images: torch.Tensor, tokens: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """images f32 [B, 3, S, S] (preprocessed).  tokens=False -> f32 [B, out_dim] = ln_post(cls) @ proj
        (open_clip encode_image; the attention-pool head for SigLIP towers); tokens=True -> f32 [B, T, width] after
        ln_post (PE forward_features(norm=True))."""
        s = self.spec
        x = L.dev(images, torch.float32, "images")
        b = x.shape[0]
        if tuple(x.shape[1:]) != (3, s.image_size, s.image_size):
            raise L.OvoHipError(f"expected [B, 3, {s.image_size}, {s.image_size}], got {tuple(x.shape)}")
        cfg = self._config(0 if tokens else (2 if s.map_pool else 1))
        if out is None:
            shape = (b, s.tokens, s.width) if tokens else (b, s.out_dim)
            out = torch.empty(shape, dtype=torch.float32, device=x.device)
        lib = L.load()
        if self.split_streams and b >= 2:
            return self._forward_split(lib, cfg, x, out)
        ws, need = self._workspace(cfg, b)
        L.check(lib.ovo_vit_forward(C.byref(cfg), C.byref(self._weights), L.ptr(x), b, L.ptr(out), L.ptr(ws), need, L.stream()))
        return out

    def _forward_split(self, lib, cfg, x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
        """The batch as two independent layer chains on two HIP streams (the caller's and a side stream), each with its own
        workspace: a chain is a strict sequence of ~200 small launches, so two of them interleave their LayerNorm / RoPE /
        attention phases with each other's GEMMs."""
        b = x.shape[0]
        h0 = (b + 1) // 2
        main = torch.cuda.current_stream(x.device)
        if self._side is None:
            self._side = torch.cuda.Stream(device=x.device)
        need0 = lib.ovo_vit_workspace_bytes(C.byref(cfg), h0)
        need1 = lib.ovo_vit_workspace_bytes(C.byref(cfg), b - h0)
        if self._ws2 is None or self._ws2[0].numel() < need0 or self._ws2[1].numel() < need1:
            self._ws2 = (torch.empty(need0, dtype=torch.uint8, device=x.device), torch.empty(max(need1, 1), dtype=torch.uint8, device=x.device))
        self._side.wait_stream(main)                          # inputs (and the previous readers of `out`) are ordered on `main`
        L.check(lib.ovo_vit_forward(C.byref(cfg), C.byref(self._weights), L.ptr(x[:h0]), h0, L.ptr(out[:h0]), L.ptr(self._ws2[0]), need0, L.stream()))
        with torch.cuda.stream(self._side):
            L.check(lib.ovo_vit_forward(C.byref(cfg), C.byref(self._weights), L.ptr(x[h0:]), b - h0, L.ptr(out[h0:]), L.ptr(self._ws2[1]), need1,
                                        L.stream()))
        main.wait_stream(self._side)
        return out

    encode_image = forward
