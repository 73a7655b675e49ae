"""TextRegion-style region pooling on PE patch tokens, on MI355X.

Mirror of the reference's `ovo/entities/textregion.py:PETextRegion` (same constructor arguments and methods).
What runs where:
  get_img_features    (textregion.py:104-143)  crop list on the host, `ovo_resize_normalize` per crop, ONE batched
                                               `ovo_vit_forward` (tokens after ln_post)
  get_features_mask   (:145-161)               `ovo_feature_masks`: bilinear mask resample -> {0,1} token weights + counts
  resize_features     (:9-28)                  `ovo_stitch_tokens_t`: 0.5 * upsampled global grid + tile tokens
  pe_value_with_sam2_attn (:163-195)           the reference repeats the tokens N times and runs a full
        nn.MultiheadAttention whose keys are all identical, i.e. a uniform average over the un-masked tokens.
        Restated exactly (identity pinned by tests/golden/textregion.npz) as
            masked mean = (W . X) / count                       -> one [N x T] . [T x d] MFMA GEMM
            out = L2( mean . (W_v^T W_o^T proj) + (b_v W_o^T + b_o) proj )   -> one pre-folded [d x d] GEMM
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import os

import numpy as np
import torch

from .. import _lib as L
from ..encoders.vit import HipViT


def _gemm(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor) -> torch.Tensor:
    """out f32[M, N] = a bf16[M, K] @ w bf16[N, K]^T + bias."""
    g = L.Gemm()
    g.A, g.lda, g.W, g.ldw = a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0)
    g.bias = bias.data_ptr() if bias is not None else None
    g.C, g.ldc, g.add, g.ld_add = out.data_ptr(), out.stride(0), None, 0
    g.M, g.N, g.K = a.shape[0], w.shape[0], a.shape[1]
    g.in_dtype, g.out_dtype, g.act, g.alpha = 2, 0, 0, 1.0
    L.check(L.load().ovo_gemm(L.C.byref(g), L.stream()))
    return out


class PETextRegion(torch.nn.Module):
    def __init__(self, model: HipViT, model_card: str = "PE-Core-L14-336", preprocess=None,
                 resize_method: str = "multi_resolution", remove_global_patch: bool = True,
                 global_patch_threshold: float = 0.07, crop_size: Optional[int] = None, upsample_times: int = 1,
                 mask_type: str = "soft", dtype: str = "bf16", device=None, project_and_normalize: bool = True,
                 share_identical_crops: Optional[bool] = None):
        super().__init__()
        # MI355X extension, OFF by default (the reference forwards every crop): on frames smaller than two crop sizes per side -- 640 x 480
        # with 336-pixel crops -- the tiling's one tile IS the global image (textregion.py:104-143: crops = [whole, whole]), and the
        # reference pushes the same pixels through the ViT twice.  With this switch identical crop rectangles are encoded once and their
        # tokens used for both; the descriptors are the same bits (test_textregion_shared_crops_equal_separate_forwards).
        self.share_identical_crops = bool(int(os.environ.get("OVO_SHARE_CROPS", "0"))) if share_identical_crops is None else bool(share_identical_crops)
        self._uniq, self._crop_index = [0], [0]
        if not model_card.startswith("PE"):
            raise NotImplementedError("Current TextRegion implementaion supports only PE models.")
        self.remove_global_patch = bool(remove_global_patch)
        self.global_patch_threshold = float(global_patch_threshold)
        if upsample_times != 1 or mask_type != "soft":
            raise NotImplementedError("only upsample_times=1 / mask_type='soft' (the reference defaults)")
        self.vlm = model
        self.model_card = model_card
        self.resize_method = resize_method
        self.project_and_normalize = project_and_normalize
        self.patch_size = model.spec.patch
        self.crop_size = crop_size if crop_size is not None else model.spec.image_size
        if self.crop_size != model.spec.image_size:
            raise L.OvoHipError("crop_size must equal the encoder's image size")
        self.device = model.device
        pw = model.pool_weights
        if pw is None:
            raise L.OvoHipError("the encoder has no attention-pool weights (PE models only)")
        d = model.spec.width
        wv, bv = pw["attn.in_proj_weight"][2 * d:].double(), pw["attn.in_proj_bias"][2 * d:].double()
        wo, bo = pw["attn.out_proj.weight"].double(), pw["attn.out_proj.bias"].double()
        mat, vec = wo @ wv, bv @ wo.T + bo                        # v -> out_proj, still d-dimensional
        if project_and_normalize:
            proj = model.proj.double().cpu()
            mat, vec = proj.T @ mat, vec @ proj                    # [D_out, d], [D_out]
        self._fold_w = mat.to(self.device, torch.bfloat16).contiguous()
        self._fold_b = vec.to(self.device, torch.float32).contiguous()
        self.out_dim = self._fold_w.shape[0]

    # ------------------------------------------------------------------ tiling (textregion.py:104-143)
    def _crops(self, h: int, w: int) -> List[Tuple[int, int, int, int]]:
        if self.resize_method != "multi_resolution":
            self.crop_num_h = self.crop_num_w = 1
            self.points_per_h = self.points_per_w = self.crop_size // self.patch_size
            self._uniq, self._crop_index = [0], [0]
            return [(0, 0, h, w)]
        nh, nw = max(h // self.crop_size, 1), max(w // self.crop_size, 1)
        self.crop_num_h, self.crop_num_w = nh, nw
        self.points_per_h = (self.crop_size // self.patch_size) * nh
        self.points_per_w = (self.crop_size // self.patch_size) * nw
        ch, cw = int(np.ceil(h / nh)), int(np.ceil(w / nw))
        crops = [(0, 0, h, w)]
        for i in range(nh):
            for j in range(nw):
                y2, x2 = min(i * ch + ch, h), min(j * cw + cw, w)
                y1, x1 = max(y2 - ch, 0), max(x2 - cw, 0)
                crops.append((y1, x1, y2 - y1, x2 - x1))
        self._uniq, self._crop_index = [], []                    # crop k uses the tokens of forwarded crop _crop_index[k] (= k without sharing)
        for k, c in enumerate(crops):
            first = crops.index(c) if getattr(self, "share_identical_crops", False) else k
            if first == k:
                self._uniq.append(k)
            self._crop_index.append(self._uniq.index(first))
        return crops

    def forward_crops(self, h: int, w: int) -> List[Tuple[int, int, int, int]]:
        """The crops that go through the encoder: all of `_crops` -- or, with `share_identical_crops`, each distinct rectangle once."""
        crops = self._crops(h, w)
        return [crops[k] for k in self._uniq]

    def get_img_features(self, image: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
        """image [3, H, W] (f32 in [0,1], or u8 with scale = 1/255) -> f32 [crops, 1 + P*P, width] after ln_post."""
        _, h, w = image.shape
        batch = self.vlm.preprocess(image, self.forward_crops(h, w), scale=scale)
        return self.vlm.forward(batch, tokens=True)

    def get_features_mask(self, region_masks: torch.Tensor):
        """bool [N, H, W] -> (bf16 [N, gpad] {0,1} token weights, f32 [N] counts)."""
        m = region_masks.view(torch.uint8) if region_masks.dtype == torch.bool else \
            (region_masks if region_masks.dtype == torch.uint8 else region_masks.to(torch.uint8))       # bool -> u8: the same bytes, no copy
        m = L.dev(m.contiguous(), torch.uint8, "region_masks")
        n, h, w = m.shape
        g = self.points_per_h * self.points_per_w
        gpad = (g + 31) // 32 * 32
        weights = torch.empty((n, gpad), dtype=torch.bfloat16, device=m.device)
        cnt = torch.empty(n, dtype=torch.float32, device=m.device)
        L.check(L.load().ovo_feature_masks(L.ptr(m), n, h, w, self.points_per_h, self.points_per_w, L.ptr(weights), gpad,
                                           L.ptr(cnt), L.stream()))
        return weights, cnt

    def pe_value_with_sam2_attn(self, feature_masks, input_feature: torch.Tensor) -> torch.Tensor:
        weights, cnt = feature_masks
        tok = L.dev(input_feature, torch.float32, "input_feature")
        if tok.shape[0] == len(self._uniq) and len(self._uniq) != len(self._crop_index):     # shared crops: one token block serves several crops
            tok = tok.index_select(0, torch.tensor(self._crop_index, dtype=torch.int64).to(tok.device, non_blocking=True))
        ncrop, tpc, d = tok.shape
        p = self.crop_size // self.patch_size
        t0 = 1 if self.vlm.spec.cls_token else 0
        nh, nw = (self.crop_num_h, self.crop_num_w) if self.resize_method == "multi_resolution" else (1, 1)
        if ncrop != 1 + nh * nw and self.resize_method == "multi_resolution":
            raise L.OvoHipError(f"expected {1 + nh * nw} crops, got {ncrop}")
        n, gpad = weights.shape
        lib = L.load()
        x_t = torch.empty((d, gpad), dtype=torch.bfloat16, device=tok.device)
        if self.resize_method == "multi_resolution":
            L.check(lib.ovo_stitch_tokens_t(L.ptr(tok), tpc, t0, d, p, nh, nw, L.ptr(x_t), gpad, L.stream()))
        else:                                                     # single crop: x_input = tokens (no 0.5*global term)
            x_t.zero_()
            x_t[:, :p * p] = tok[0, t0:].t().to(torch.bfloat16)
        if self.remove_global_patch and n > 0:
            weights, cnt = self._remove_global_patch(x_t, weights, cnt, nh * nw * p * p if self.resize_method == "multi_resolution" else p * p)
        sums = _gemm(weights, x_t, None, torch.empty((n, d), dtype=torch.float32, device=tok.device))
        mean = torch.empty((n, d), dtype=torch.bfloat16, device=tok.device)
        L.check(lib.ovo_scale_rows_bf16(L.ptr(sums), L.ptr(cnt), n, d, L.ptr(mean), L.stream()))
        out = _gemm(mean, self._fold_w, self._fold_b, torch.empty((n, self.out_dim), dtype=torch.float32, device=tok.device))
        if not self.project_and_normalize:
            return out
        L.check(lib.ovo_l2_normalize_rows(L.ptr(out), n, self.out_dim, L.ptr(out), L.stream()))
        return out

    def _remove_global_patch(self, x_t: torch.Tensor, weights: torch.Tensor, cnt: torch.Tensor, g: int):
        """Reference: textregion.py:31-50.  Clears "global" token columns of the {0,1} weights and recounts.  The
        reference's [T, T] patch similarity is folded away (include/ovo_hip.h, a18): two small MFMA GEMMs instead."""
        lib = L.load()
        d, gpad = x_t.shape
        n = weights.shape[0]
        u_t = torch.empty_like(x_t)
        u = torch.empty((gpad, d), dtype=torch.bfloat16, device=x_t.device)
        L.check(lib.ovo_unit_tokens(L.ptr(x_t), d, g, gpad, L.ptr(u_t), L.ptr(u), L.stream()))
        sums = _gemm(weights, u_t, None, torch.empty((n, d), dtype=torch.float32, device=x_t.device))
        mean = torch.empty((n, d), dtype=torch.bfloat16, device=x_t.device)
        L.check(lib.ovo_scale_rows_bf16(L.ptr(sums), L.ptr(cnt), n, d, L.ptr(mean), L.stream()))
        r_t = _gemm(mean, u, None, torch.empty((n, gpad), dtype=torch.float32, device=x_t.device))
        weights, cnt = weights.clone(), torch.empty_like(cnt)     # get_features_mask's result stays usable by the caller
        L.check(lib.ovo_global_patch_filter(L.ptr(r_t), L.ptr(weights), n, g, gpad, self.global_patch_threshold, L.ptr(cnt), L.stream()))
        return weights, cnt

    def predict(self, image: torch.Tensor, region_masks: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
        """Reference: textregion.py:197-203.  image [3, H, W] in [0, 1] -> f32 [N, D] unit descriptors."""
        feats = self.get_img_features(image, scale=scale)
        return self.pe_value_with_sam2_attn(self.get_features_mask(region_masks), feats)
