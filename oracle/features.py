"""Oracle: similarity / crop-descriptor fusion / TextRegion region pooling / mask NMS (numpy+torch CPU).

TEST INFRASTRUCTURE (see oracle/__init__.py).
"""
from __future__ import annotations

import ctypes
from typing import List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import lib


# ------------------------------------------------------------------ similarity / query (a21, a22)
def similarity(feats: np.ndarray, texts: np.ndarray, siglip: bool = False, logit_scale: float = 0.0,
               logit_bias: float = 0.0) -> np.ndarray:
    """clip_utils.py:10-19 -> f32[N,Q] (double accumulation, rounded once)."""
    f = np.ascontiguousarray(feats, np.float32)
    t = np.ascontiguousarray(texts, np.float32)
    out = np.empty((f.shape[0], t.shape[0]), np.float32)
    lib().orc_similarity(f.ctypes.data_as(ctypes.c_void_p), f.shape[0], t.ctypes.data_as(ctypes.c_void_p),
                         t.shape[0], f.shape[1], int(siglip), float(logit_scale), float(logit_bias),
                         out.ctypes.data_as(ctypes.c_void_p))
    return out


def text_embeddings(table_rows: np.ndarray) -> np.ndarray:
    """clip_generator.py:161-173,193-196: per query, unit-normalise each template embedding, average,
    unit-normalise again.  table_rows: f32[Q, n_templates, D] raw text-tower outputs."""
    t = torch.from_numpy(np.asarray(table_rows, np.float32))
    t = t / t.norm(dim=-1, keepdim=True)
    return F.normalize(t.mean(1), p=2, dim=-1).numpy()


def classify(sim: np.ndarray, th: float = 0.0) -> Tuple[np.ndarray, np.ndarray]:
    """ovo.py:487-491: first-max argmax, conf = max; conf <= th -> class -1, conf 0."""
    cls = sim.argmax(1).astype(np.int64)
    conf = sim[np.arange(sim.shape[0]), cls].astype(np.float32)
    low = conf <= th
    cls[low] = -1
    conf[low] = 0
    return cls, conf


def fuse_crop_descriptors(g, seg, box, mode: str, w_masked: float, w_global: float) -> np.ndarray:
    """clip_utils.py:21-48 fuse_clips."""
    g, seg, box = (torch.from_numpy(np.asarray(a, np.float32)) for a in (g, seg, box))

    def cos(a, b):
        return F.cosine_similarity(a, b, dim=-1, eps=1e-6)
    if mode in ("hovsg", "fixed_weights"):
        loc = F.normalize(seg * w_masked + box * (1 - w_masked), p=2, dim=-1)
        wg = w_global if mode == "fixed_weights" else torch.softmax(cos(g, loc), dim=0).unsqueeze(1)
        out = F.normalize(g * wg + loc * (1 - wg), p=2, dim=-1)
    elif mode == "adaptive_weights":
        wl = (cos(seg, box) * w_masked).unsqueeze(-1)
        loc = F.normalize(seg * wl + box * (1 - wl), p=2, dim=-1)
        wg = (cos(g, loc) * w_global).unsqueeze(-1)
        out = F.normalize(g * wg + loc * (1 - wg), p=2, dim=-1)
    elif mode == "concept_fusion":
        wg = torch.softmax(cos(g, box), dim=0).unsqueeze(-1)
        out = F.normalize(wg * g + (1 - wg) * box, p=2, dim=-1)
    else:
        out = seg
    return out.numpy()


# ------------------------------------------------------------------ TextRegion pooling (a15-a17)
def feature_masks(masks: np.ndarray, gh: int, gw: int) -> np.ndarray:
    """textregion.py:145-161: bilinear (align_corners=False) resample of bool masks to the token grid,
    clamped to [0,1] -> f32[N, gh*gw]."""
    m = torch.from_numpy(np.asarray(masks)).float()[None]
    out = F.interpolate(m, [gh, gw], mode="bilinear")
    return out.reshape(-1, gh * gw).clamp(0, 1).numpy()


def stitch_tokens(tokens: np.ndarray, P: int, gh: int, gw: int, nh: int, nw: int) -> np.ndarray:
    """textregion.py:9-28 resize_features: tokens f32[1+nh*nw, P*P, D] (cls already dropped) ->
    f32[gh*gw, D]: bilinear up-sampled global grid, then per tile 0.5*global + tile."""
    t = torch.from_numpy(np.asarray(tokens, np.float32))
    b, _, d = t.shape
    grid = t.permute(0, 2, 1).reshape(b, d, P, P)
    out = F.interpolate(grid[:1], [gh, gw], mode="bilinear")
    k = 1
    for i in range(nh):
        for j in range(nw):
            ys, xs = slice(i * P, (i + 1) * P), slice(j * P, (j + 1) * P)
            out[:, :, ys, xs] = 0.5 * out[:, :, ys, xs] + grid[k]
            k += 1
    return out.reshape(d, gh * gw).T.contiguous().numpy()


def region_pool(x: np.ndarray, fmask: np.ndarray, w_v, b_v, w_o, b_o, proj, normalize: bool = True) -> np.ndarray:
    """textregion.py:163-195 pe_value_with_sam2_attn, restated.

    Every key is the same vector, so the attention weights are uniform over the un-masked tokens:
    out = ((mean_{mask>0} x) W_v^T + b_v) W_o^T + b_o, then @ proj and L2-normalise
    (identity verified against the reference's nn.MultiheadAttention path via tests/golden/textregion.npz).
    x f32[T,D], fmask f32[N,T] -> f32[N,D_out].  A mask with no token gives NaN in the reference
    (softmax over an all-masked row); here it gives NaN too (0/0)."""
    x = torch.from_numpy(np.asarray(x, np.float64))
    sel = torch.from_numpy((np.asarray(fmask) > 0).astype(np.float64))
    mean = (sel @ x) / sel.sum(1, keepdim=True)
    v = mean @ torch.from_numpy(np.asarray(w_v, np.float64)).T + torch.from_numpy(np.asarray(b_v, np.float64))
    o = v @ torch.from_numpy(np.asarray(w_o, np.float64)).T + torch.from_numpy(np.asarray(b_o, np.float64))
    if not normalize:
        return o.float().numpy()
    r = o @ torch.from_numpy(np.asarray(proj, np.float64))
    return F.normalize(r, dim=-1).float().numpy()


# ------------------------------------------------------------------ mask NMS / seg map (a11)
def pack_masks(masks: np.ndarray) -> Tuple[np.ndarray, int]:
    n = masks.shape[0]
    flat = masks.reshape(n, -1)
    pad = (-flat.shape[1]) % 64
    if pad:
        flat = np.concatenate([flat, np.zeros((n, pad), bool)], 1)
    bits = np.packbits(flat, axis=1, bitorder="little").view(np.uint64)
    return np.ascontiguousarray(bits), bits.shape[1]


def mask_intersections(masks: np.ndarray) -> np.ndarray:
    bits, words = pack_masks(masks)
    n = masks.shape[0]
    inter = np.zeros((n, n), np.int32)
    lib().orc_mask_intersections(bits.ctypes.data_as(ctypes.c_void_p), n, words, inter.ctypes.data_as(ctypes.c_void_p))
    return inter


def mask_nms(masks: np.ndarray, scores: np.ndarray, iou_thr=0.8, score_thr=0.7, inner_thr=0.5) -> np.ndarray:
    """segment_utils.py:195-259 mask_nms -> kept indices (into `masks`), in descending-score order.

    Counts are exact integers; ratios are formed in fp32 from fp32-converted counts like the
    reference (torch.sum(..., dtype=float))."""
    s = torch.from_numpy(np.asarray(scores, np.float32))
    s_sorted, order = s.sort(0, descending=True)
    order = order.numpy()
    inter = mask_intersections(masks[order]).astype(np.float32)
    n = len(order)
    area = np.diag(inter).copy()
    iou = np.zeros((n, n), np.float32)
    inner = np.zeros((n, n), np.float32)
    for i in range(n):
        for j in range(i, n):
            it = inter[i, j]
            union = area[i] + area[j] - it
            iou[i, j] = it / union
            ri, rj = it / area[i], it / area[j]
            if ri < 0.5 and rj >= 0.85:
                inner[i, j] = np.float32(1) - rj * ri
            if ri >= 0.85 and rj < 0.5:
                inner[j, i] = np.float32(1) - rj * ri
    iou_max = np.triu(iou, 1).max(0)
    in_u = np.triu(inner, 1).max(0)
    in_l = np.tril(inner, 1).max(0)          # diagonal=1 oddity kept (segment_utils.py:237)
    keep = iou_max <= np.float32(iou_thr)
    conf = s_sorted.numpy() > np.float32(score_thr)
    ku = in_u <= np.float32(1 - inner_thr)
    kl = in_l <= np.float32(1 - inner_thr)
    top3 = np.argsort(-s_sorted.numpy(), kind="stable")[:3]
    for arr in (conf, ku, kl):
        if arr.sum() == 0:
            arr[top3] = True
    keep = keep & conf & ku & kl
    return order[keep]


def paint_segmap(masks: np.ndarray, stability: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """segment_utils.py:12-27 mask2segmap(sort=True): masks ordered by descending stability
    (heapq.nlargest is stable for ties), earlier masks win overlaps."""
    order = sorted(range(len(stability)), key=lambda i: -float(stability[i]))
    m = masks[order]
    seg = np.full(masks.shape[1:], -1, np.int32)
    for i, mk in enumerate(m):
        seg[mk & (seg < 0)] = i
    return seg, m


def masks_to_boxes(masks: np.ndarray) -> np.ndarray:
    """segment_utils.py:43-94 batched_mask_to_box -> i64[N,4] xyxy, zeros for an empty mask."""
    out = np.zeros((masks.shape[0], 4), np.int64)
    for i, m in enumerate(masks):
        ys, xs = np.nonzero(m)
        if ys.size:
            out[i] = (xs.min(), ys.min(), xs.max(), ys.max())
    return out


def mask_crops(masks: np.ndarray, image: np.ndarray, also_bbox: bool, margin: int = 50, out_l: int = 224) -> np.ndarray:
    """segment_utils.py:29-41 segmap2segimg with its helpers (:88-94 xyxy->xywh over inclusive edges, :118-126
    seg_img_from_image, :128-139 get_seg_img / get_bbox_img, :141-150 pad_img, :152-172 increase_bbox_by_margin).
    masks bool [N,H,W], image [3,H,W] u8 or f32 (0..255) -> f32 [N, 3|6, out_l, out_l].
    torchvision is not installed here: F.resize on a tensor is restated as what it dispatches to,
    torch.nn.functional.interpolate(mode="bilinear", antialias=True, align_corners=False), computed in f32 and -- for a
    uint8 image -- rounded (torch.round, half to even) like torchvision's _cast_squeeze_out.  Degenerate boxes (w or h = 0,
    where the reference raises inside F.resize) give zeros."""
    import torch
    img = torch.from_numpy(np.ascontiguousarray(image))
    is_u8 = img.dtype == torch.uint8

    def resize(t: torch.Tensor) -> torch.Tensor:
        if t.shape[-1] == 0 or t.shape[-2] == 0:
            return torch.zeros((t.shape[0], out_l, out_l), dtype=torch.float32)
        r = torch.nn.functional.interpolate(t[None].float(), size=(out_l, out_l), mode="bilinear", antialias=True, align_corners=False)[0]
        return torch.round(r) if is_u8 else r

    boxes = masks_to_boxes(masks)
    out = []
    for mk, (x1, y1, x2, y2) in zip(masks, boxes):
        x, y, w, h = int(x1), int(y1), int(x2 - x1), int(y2 - y1)
        m = torch.from_numpy(np.ascontiguousarray(mk[y:y + h, x:x + w]))
        seg = torch.zeros((3, h, w), dtype=img.dtype)
        seg[:, m] = img[:, y:y + h, x:x + w][:, m]
        if also_bbox:
            bx, by, bw, bh = x - margin, y - margin, w + 2 * margin, h + 2 * margin
            if bx < 0:
                bw, bx = bw + bx, 0
            if by < 0:
                bh, by = bh + by, 0
            box = img[:, by:by + max(bh, 0), bx:bx + max(bw, 0)]
            out.append(torch.cat([resize(seg), resize(box)], 0))
        else:
            side = max(w, h)
            pad = torch.zeros((3, side, side), dtype=img.dtype)
            if h > w:
                pad[..., (h - w) // 2:(h - w) // 2 + w] = seg
            else:
                pad[:, (w - h) // 2:(w - h) // 2 + h, :] = seg
            out.append(resize(pad))
    if not out:
        return np.zeros((0, 6 if also_bbox else 3, out_l, out_l), np.float32)
    return torch.stack(out).numpy().astype(np.float32)


def remove_global_patch(x: np.ndarray, fmask: np.ndarray, th: float = 0.07) -> Tuple[np.ndarray, np.ndarray]:
    """textregion.py:31-50, literally (with the [T, T] patch similarity).  x f32 [T, d] stitched tokens, fmask f32 [N, T]
    -> (fmask with the "global" token columns cleared, the per-token difference score f32 [T])."""
    import torch
    xi = torch.from_numpy(np.asarray(x, np.float32))[None]
    fm = torch.from_numpy(np.asarray(fmask, np.float32)).clone()
    pf = (xi / xi.norm(dim=-1, keepdim=True))[0]
    sim = pf @ pf.T
    p2r = sim @ (fm > 0).float().T
    avg = p2r / (fm > 0).sum(dim=-1)
    belong = (avg * (fm > 0).float().T).sum(dim=-1) / ((fm > 0).sum(dim=0) + 1e-9)
    outside = (avg * (fm == 0).float().T).sum(dim=-1) / ((fm == 0).sum(dim=0) + 1e-9)
    diff = (belong - outside).float().numpy()
    fm[:, torch.from_numpy(diff < th)] = 0
    return fm.numpy(), diff
