"""SAM2 image encoder (Hiera trunk + FPN neck) on MI355X.

Replaces the encoder half of `SAM2AutomaticMaskGenerator.generate(image)` that the reference calls through the
un-vendored `sam2` package (mask_generator.py:113, segment_utils.py:291-308) with `ovo_hiera_forward` from
libovo_hip.so.  Weights use the sam2 repository's `image_encoder.*` state-dict names (trunk.* / neck.*) plus
`sam_mask_decoder.conv_s0/1`, so a real SAM2.1 checkpoint loads unchanged; offline, `random_state` gives seeded
weights of the same architecture.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from .. import _lib as L

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


@dataclass(frozen=True)
class HieraSpec:
    name: str
    embed_dim: int
    num_heads: int
    stages: Tuple[int, int, int, int]
    global_blocks: Tuple[int, ...]
    window_spec: Tuple[int, int, int, int]
    pos_bkg: Tuple[int, int]
    image_size: int = 1024
    fpn_dim: int = 256
    hi_res: bool = True

    @property
    def dims(self):
        return tuple(self.embed_dim * 2 ** i for i in range(4))

    @property
    def heads(self):
        return tuple(self.num_heads * 2 ** i for i in range(4))

    def flops_per_image(self) -> float:
        """Dense FLOPs of one forward, counted the way the kernels run (padding windows included)."""
        s4 = self.image_size // 4
        f = 2.0 * s4 * s4 * 147 * self.embed_dim
        h, idx = s4, 0
        for s, nb in enumerate(self.stages):
            for b in range(nb):
                first = s > 0 and b == 0
                din, dout = (self.dims[s - 1] if first else self.dims[s]), self.dims[s]
                ws = self.window_spec[s - 1] if first else self.window_spec[s]
                if idx in self.global_blocks:
                    ws = 0
                nw = 1 if ws == 0 else (-(-h // ws)) ** 2        # windows, padding ones included
                tk = h * h if ws == 0 else ws * ws
                rows = nw * tk
                tq = tk // 4 if first else tk
                ho = h // 2 if first else h
                f += 2.0 * rows * din * 3 * dout + (2.0 * rows * din * dout if first else 0)
                f += 4.0 * nw * tq * tk * dout + 2.0 * nw * tq * dout * dout
                f += 16.0 * ho * ho * dout * dout
                h, idx = ho, idx + 1
            f += 2.0 * h * h * self.dims[s] * self.fpn_dim
        if self.hi_res:
            f += 2.0 * s4 * s4 * self.fpn_dim * 32 + 2.0 * (s4 // 2) ** 2 * self.fpn_dim * 64
        return f


# sam2 configs (sam2/configs/sam2.1/*.yaml); the reference's table has hiera_l and hiera_t (segment_utils.py:274),
# BASELINE.json's "SAM2-base" is hiera_b+
SPECS: Dict[str, HieraSpec] = {
    "hiera_t": HieraSpec("hiera_t", 96, 1, (1, 2, 7, 2), (5, 7, 9), (8, 4, 14, 7), (7, 7)),
    "hiera_s": HieraSpec("hiera_s", 96, 1, (1, 2, 11, 2), (7, 10, 13), (8, 4, 14, 7), (7, 7)),
    "hiera_b+": HieraSpec("hiera_b+", 112, 2, (2, 3, 16, 3), (12, 16, 20), (8, 4, 14, 7), (14, 14)),
    "hiera_l": HieraSpec("hiera_l", 144, 2, (2, 6, 36, 4), (23, 33, 43), (8, 4, 16, 8), (7, 7)),
    "hiera_test": HieraSpec("hiera_test", 32, 1, (1, 2, 3, 2), (4,), (8, 4, 6, 4), (5, 5), image_size=256, fpn_dim=64),
    # small trunk with the real FPN width: pairs with the SAM2 mask decoder (sam_decoder.SPECS["sam2_small"]) in tests
    "hiera_test256": HieraSpec("hiera_test256", 32, 1, (1, 2, 3, 2), (4,), (8, 4, 6, 4), (5, 5), image_size=256, fpn_dim=256),
}


def block_plan(spec: HieraSpec) -> List[Tuple[int, int]]:
    """(dim_in, dim_out) of every block, in order."""
    out = []
    for s, nb in enumerate(spec.stages):
        for b in range(nb):
            out.append((spec.dims[s - 1] if (s > 0 and b == 0) else spec.dims[s], spec.dims[s]))
    return out


def random_state(spec: HieraSpec, seed: int = 0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)

    def rn(*shape, std=0.02):
        return torch.randn(*shape, generator=g) * std
    e = spec.embed_dim
    sd = {"trunk.patch_embed.proj.weight": rn(e, 3, 7, 7, std=0.05), "trunk.patch_embed.proj.bias": rn(e),
          "trunk.pos_embed": rn(1, e, *spec.pos_bkg), "trunk.pos_embed_window": rn(1, e, spec.window_spec[0], spec.window_spec[0])}
    for i, (din, dout) in enumerate(block_plan(spec)):
        p = f"trunk.blocks.{i}."
        sd[p + "norm1.weight"], sd[p + "norm1.bias"] = 1 + rn(din, std=0.05), rn(din, std=0.05)
        sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"] = rn(3 * dout, din, std=din ** -0.5), rn(3 * dout)
        sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"] = rn(dout, dout, std=dout ** -0.5), rn(dout)
        sd[p + "norm2.weight"], sd[p + "norm2.bias"] = 1 + rn(dout, std=0.05), rn(dout, std=0.05)
        sd[p + "mlp.layers.0.weight"], sd[p + "mlp.layers.0.bias"] = rn(4 * dout, dout, std=dout ** -0.5), rn(4 * dout)
        sd[p + "mlp.layers.1.weight"], sd[p + "mlp.layers.1.bias"] = rn(dout, 4 * dout, std=(4 * dout) ** -0.5), rn(dout)
        if din != dout:
            sd[p + "proj.weight"], sd[p + "proj.bias"] = rn(dout, din, std=din ** -0.5), rn(dout)
    for j in range(4):                                          # neck.convs[0] is the COARSEST level
        c = spec.dims[3 - j]
        sd[f"neck.convs.{j}.conv.weight"], sd[f"neck.convs.{j}.conv.bias"] = rn(spec.fpn_dim, c, 1, 1, std=c ** -0.5), rn(spec.fpn_dim)
    sd["sam_mask_decoder.conv_s0.weight"], sd["sam_mask_decoder.conv_s0.bias"] = rn(32, spec.fpn_dim, 1, 1, std=spec.fpn_dim ** -0.5), rn(32)
    sd["sam_mask_decoder.conv_s1.weight"], sd["sam_mask_decoder.conv_s1.bias"] = rn(64, spec.fpn_dim, 1, 1, std=spec.fpn_dim ** -0.5), rn(64)
    return sd


def position_embedding(spec: HieraSpec, sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    """Hiera's windowed position embedding: bicubic-resized background + tiled window embedding -> [(S/4)^2, C]."""
    s4 = spec.image_size // 4
    bkg = F.interpolate(sd["trunk.pos_embed"].float(), size=(s4, s4), mode="bicubic")
    win = sd["trunk.pos_embed_window"].float()
    pos = bkg + win.tile([1, 1, s4 // win.shape[2], s4 // win.shape[3]])
    return pos[0].permute(1, 2, 0).reshape(s4 * s4, -1).contiguous()


def _pad_k(w: torch.Tensor) -> torch.Tensor:
    k = w.shape[1]
    kp = (k + 63) // 64 * 64                       # the rule of hiera.hip (padk)
    if kp == k:
        return w
    out = torch.zeros(w.shape[0], kp, dtype=w.dtype)
    out[:, :k] = w
    return out


class HipHiera:
    def __init__(self, spec: HieraSpec, state: Optional[Dict[str, torch.Tensor]] = None, device="cuda", seed: int = 0):
        self.spec, self.device = spec, torch.device(device)
        sd = state if state is not None else random_state(spec, seed)
        sd = {(k[len("image_encoder."):] if k.startswith("image_encoder.") else k): v for k, v in sd.items()}
        self._keep: List[torch.Tensor] = []

        def mat(t):
            x = _pad_k(t.detach().float().reshape(t.shape[0], -1)).to(self.device, torch.bfloat16).contiguous()
            self._keep.append(x)
            return x.data_ptr()

        def vec(t):
            x = t.detach().to(self.device, torch.float32).contiguous()
            self._keep.append(x)
            return x.data_ptr()

        plan = block_plan(spec)
        self.q_prescaled = L.q_prescale_enabled()
        stage_of = [s for s, nb in enumerate(spec.stages) for _ in range(nb)]
        self._blocks = (L.HieraBlock * len(plan))()
        for i, (din, dout) in enumerate(plan):
            p, b = f"trunk.blocks.{i}.", self._blocks[i]
            b.ln1_g, b.ln1_b = vec(sd[p + "norm1.weight"]), vec(sd[p + "norm1.bias"])
            qw, qb = sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]
            if self.q_prescaled:                           # log2 e / sqrt(hd) into the q rows (the 2 x 2 max-pool of q commutes with a positive factor)
                qw, qb = L.fold_q_scale(qw, qb, dout, dout // spec.heads[stage_of[i]])
            b.qkv_w, b.qkv_b = mat(qw), vec(qb)
            b.out_w, b.out_b = mat(sd[p + "attn.proj.weight"]), vec(sd[p + "attn.proj.bias"])
            b.ln2_g, b.ln2_b = vec(sd[p + "norm2.weight"]), vec(sd[p + "norm2.bias"])
            b.fc1_w, b.fc1_b = mat(sd[p + "mlp.layers.0.weight"]), vec(sd[p + "mlp.layers.0.bias"])
            b.fc2_w, b.fc2_b = mat(sd[p + "mlp.layers.1.weight"]), vec(sd[p + "mlp.layers.1.bias"])
            if din != dout:
                b.res_w, b.res_b = mat(sd[p + "proj.weight"]), vec(sd[p + "proj.bias"])
        w = L.HieraWeights()
        conv = sd["trunk.patch_embed.proj.weight"].float().reshape(spec.embed_dim, -1)
        pw = torch.zeros(spec.embed_dim, 192)
        pw[:, :147] = conv
        w.patch_w, w.patch_b = mat(pw), vec(sd["trunk.patch_embed.proj.bias"])
        w.pos = vec(position_embedding(spec, sd))
        w.blocks = C.cast(self._blocks, C.POINTER(L.HieraBlock))
        for s in range(4):                                      # level s (fine -> coarse) = neck.convs[3 - s]
            w.neck_w[s] = mat(sd[f"neck.convs.{3 - s}.conv.weight"])
            w.neck_b[s] = vec(sd[f"neck.convs.{3 - s}.conv.bias"])
        self.hi_res = spec.hi_res and "sam_mask_decoder.conv_s0.weight" in sd
        if self.hi_res:
            w.s0_w, w.s0_b = mat(sd["sam_mask_decoder.conv_s0.weight"]), vec(sd["sam_mask_decoder.conv_s0.bias"])
            w.s1_w, w.s1_b = mat(sd["sam_mask_decoder.conv_s1.weight"]), vec(sd["sam_mask_decoder.conv_s1.bias"])
        self._weights = w
        cfg = L.HieraConfig()
        cfg.image_size = spec.image_size
        cfg.dims[:], cfg.heads[:], cfg.blocks[:], cfg.window[:] = spec.dims, spec.heads, spec.stages, spec.window_spec
        cfg.n_global = len(spec.global_blocks)
        for i, gb in enumerate(spec.global_blocks):
            cfg.global_blocks[i] = gb
        cfg.fpn_dim, cfg.hi_res, cfg.ln_eps = spec.fpn_dim, int(self.hi_res), 1e-6
        cfg.q_prescaled = int(self.q_prescaled)
        self._cfg = cfg
        self._ws: Optional[torch.Tensor] = None

    # ---------------------------------------------------------------- preprocessing (SAM2Transforms: Resize + Normalize)
    def preprocess(self, image: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """CHW u8 (0..255) or f32 ([0,1]) image on the GPU, or an HWC u8 frame (read in place) -> f32 [1, 3, S, S]."""
        s = self.spec.image_size
        img = L.dev(image, image.dtype, "image")
        hwc = img.dtype == torch.uint8 and img.shape[-1] == 3 and img.shape[0] != 3
        h, w = (img.shape[0], img.shape[1]) if hwc else (img.shape[1], img.shape[2])
        if out is None:
            out = torch.empty((1, 3, s, s), dtype=torch.float32, device=img.device)
        mean, std = (C.c_float * 3)(*IMAGENET_MEAN), (C.c_float * 3)(*IMAGENET_STD)
        scale = 1.0 / 255.0 if img.dtype == torch.uint8 else 1.0
        L.check(L.load().ovo_resize_normalize(L.ptr(img), 4 if hwc else L.DTYPE_CODE[img.dtype], 3, h, w, 0, 0, h, w, L.ptr(out), s, s, 1, scale,
                                              mean, std, L.stream()))
        return out

    def preprocess_batch(self, images, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`preprocess` of several frames -> f32 [len(images), 3, S, S]; frames of one size / layout as ONE launch (`ovo_resize_normalize_batch`)."""
        s = self.spec.image_size
        imgs = [L.dev(im, im.dtype, "image") for im in images]
        first = imgs[0]
        if out is None:
            out = torch.empty((len(imgs), 3, s, s), dtype=torch.float32, device=first.device)
        if not all(im.dtype == first.dtype and im.shape == first.shape and im.device == first.device for im in imgs) or first.dtype not in (torch.uint8, torch.float32):
            for k, im in enumerate(imgs):
                self.preprocess(im, out=out[k:k + 1])
            return out
        hwc = first.dtype == torch.uint8 and first.shape[-1] == 3 and first.shape[0] != 3
        h, w = (first.shape[0], first.shape[1]) if hwc else (first.shape[1], first.shape[2])
        mean, std = (C.c_float * 3)(*IMAGENET_MEAN), (C.c_float * 3)(*IMAGENET_STD)
        srcs = (C.c_void_p * len(imgs))(*[im.data_ptr() for im in imgs])
        rect = (C.c_int32 * 4)(0, 0, h, w)
        L.check(L.load().ovo_resize_normalize_batch(srcs, len(imgs), 4 if hwc else L.DTYPE_CODE[first.dtype], 3, h, w, rect, 1, L.ptr(out), s, s, 1,
                                                    1.0 / 255.0 if first.dtype == torch.uint8 else 1.0, mean, std, L.stream()))
        return out

    def forward(self, images: torch.Tensor):
        """images f32 [B, 3, S, S] -> (feat0 [B,S/4,S/4,c0], feat1 [B,S/8,S/8,c1], feat2 [B,S/16,S/16,fpn_dim]) NHWC f32."""
        s = self.spec
        x = L.dev(images, torch.float32, "images")
        b = x.shape[0]
        if tuple(x.shape[1:]) != (3, s.image_size, s.image_size):
            raise L.OvoHipError(f"expected [B, 3, {s.image_size}, {s.image_size}], got {tuple(x.shape)}")
        lib = L.load()
        need = lib.ovo_hiera_workspace_bytes(C.byref(self._cfg), b)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        c0, c1 = (32, 64) if self.hi_res else (s.fpn_dim, s.fpn_dim)
        s4 = s.image_size // 4
        f0 = torch.empty((b, s4, s4, c0), dtype=torch.float32, device=x.device)
        f1 = torch.empty((b, s4 // 2, s4 // 2, c1), dtype=torch.float32, device=x.device)
        f2 = torch.empty((b, s4 // 4, s4 // 4, s.fpn_dim), dtype=torch.float32, device=x.device)
        L.check(lib.ovo_hiera_forward(C.byref(self._cfg), C.byref(self._weights), L.ptr(x), b, L.ptr(f0), L.ptr(f1), L.ptr(f2),
                                      L.ptr(self._ws), need, L.stream()))
        return f0, f1, f2

    def encode_frame(self, image):
        """u8 [H, W, 3] numpy frame (or CHW device tensor) -> dict(image_embed, high_res_feats)."""
        if isinstance(image, np.ndarray):
            image = torch.from_numpy(np.ascontiguousarray(image.transpose(2, 0, 1))).to(self.device)
        f0, f1, f2 = self.forward(self.preprocess(image))
        return {"image_embed": f2, "high_res_feats": (f0, f1)}
