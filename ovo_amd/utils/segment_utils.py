"""Mask post-processing on MI355X: NMS over SAM masks, seg-map painting, mask boxes.

Mirror of the mask half of the reference's `ovo/utils/segment_utils.py` (mask_nms :195-259, masks_update
:173-186, filter :188-193, mask2segmap :12-27, batched_mask_to_box :43-94).  The reference's NMS is an
O(n^2) Python loop with two full-image reductions per pair on the CPU; here the masks are bit-packed on the
GPU and all n^2 intersections come from one popcount kernel (`ovo_mask_intersections`); the pairwise
rules then run vectorised on the tiny [n, n] table.  The SAM loader of that file (:269-309) needs the
`sam2` package and checkpoints -- see ovo_amd.encoders.hiera for the image encoder.
"""
from __future__ import annotations

import heapq
from typing import Any, Dict, List, Tuple

import numpy as np
import torch

from .. import _lib as L


def pack_masks(masks: torch.Tensor) -> Tuple[torch.Tensor, int]:
    """bool/u8 [n, H, W] on the GPU -> (u64-as-i64 [n, words], words); bit k of word j = pixel 64 j + k."""
    m = masks.reshape(masks.shape[0], -1)
    m = L.dev(m.to(torch.uint8) if m.dtype != torch.uint8 else m.contiguous(), torch.uint8, "masks")
    n, pixels = m.shape
    words = (pixels + 63) // 64
    bits = torch.empty((n, words), dtype=torch.int64, device=m.device)
    L.check(L.load().ovo_pack_masks(L.ptr(m), n, pixels, L.ptr(bits), words, L.stream()))
    return bits, words


def mask_intersections(masks: torch.Tensor) -> torch.Tensor:
    """i32[n, n] pixel counts of mask_i & mask_j (diagonal = areas)."""
    bits, words = pack_masks(masks)
    n = bits.shape[0]
    inter = torch.empty((n, n), dtype=torch.int32, device=bits.device)
    L.check(L.load().ovo_mask_intersections(L.ptr(bits), n, words, L.ptr(inter), L.stream()))
    return inter


def mask_nms(masks: torch.Tensor, scores: torch.Tensor, iou_thr: float = 0.7, score_thr: float = 0.1,
             inner_thr: float = 0.2, **kwargs) -> torch.Tensor:
    """Reference: segment_utils.py:195-259.  Returns the kept mask indices in descending-score order.

    Deviation: when NO mask passes one of the three filters the reference indexes a 1-D tensor with
    `[index, 0]` and raises IndexError (:247-255); here the intended rule (keep the top-3 scores) is applied."""
    scores_sorted, order = scores.to(masks.device).float().sort(0, descending=True)
    inter = mask_intersections(masks.to(torch.uint8).index_select(0, order)).cpu().numpy().astype(np.float32)
    s = scores_sorted.cpu().numpy()
    n = inter.shape[0]
    area = np.diag(inter).copy()
    upper = np.triu(np.ones((n, n), bool))                       # pairs (i, j) with j >= i, as in the loop
    with np.errstate(divide="ignore", invalid="ignore"):
        union = area[:, None] + area[None, :] - inter
        iou = np.where(upper, inter / union, np.float32(0)).astype(np.float32)
        ri = inter / area[:, None]                               # intersection / area[i]
        rj = inter / area[None, :]                               # intersection / area[j]
        inner_val = (np.float32(1) - rj * ri).astype(np.float32)
    inner = np.zeros((n, n), np.float32)
    a = upper & (ri < np.float32(0.5)) & (rj >= np.float32(0.85))
    inner[a] = inner_val[a]                                       # inner[i, j]
    b = upper & (ri >= np.float32(0.85)) & (rj < np.float32(0.5))
    inner.T[b] = inner_val[b]                                     # inner[j, i]
    iou_max = np.triu(iou, 1).max(0)
    inner_u = np.triu(inner, 1).max(0)
    inner_l = np.tril(inner, 1).max(0)                            # diagonal=1 kept as in the reference (:237)
    keep = iou_max <= np.float32(iou_thr)
    rules = [s > np.float32(score_thr), inner_u <= np.float32(1 - inner_thr), inner_l <= np.float32(1 - inner_thr)]
    for r in rules:
        if r.sum() == 0:
            r[np.argsort(-s, kind="stable")[:3]] = True
        keep = keep & r
    return order[torch.from_numpy(keep).to(order.device)]


def filter(keep: torch.Tensor, masks_result) -> List[Dict[str, Any]]:
    """Reference: segment_utils.py:188-193 -- kept masks in their ORIGINAL order."""
    wanted = set(keep.int().cpu().numpy().tolist())
    return [m for i, m in enumerate(masks_result) if i in wanted]


def masks_update(*args, device="cuda", **kwargs) -> Tuple[List[Dict[str, Any]], ...]:
    """Reference: segment_utils.py:173-186.  Each argument is a list of SAM mask dicts."""
    out = ()
    for level in args:
        seg = torch.from_numpy(np.stack([m["segmentation"] for m in level], axis=0)).to(device)
        iou = torch.from_numpy(np.stack([m["predicted_iou"] for m in level], axis=0))
        stab = torch.from_numpy(np.stack([m["stability_score"] for m in level], axis=0))
        out += (filter(mask_nms(seg, stab * iou, **kwargs), level),)
    return out


def masks_update_device(seg: torch.Tensor, predicted_iou, stability_score, **kwargs) -> torch.Tensor:
    """`masks_update` (segment_utils.py:173-186) for masks that are already on the GPU: seg u8/bool [n, H, W] ->
    kept indices in their ORIGINAL order (what `filter` produces), as an i64 CPU tensor."""
    scores = torch.as_tensor(np.asarray(stability_score)) * torch.as_tensor(np.asarray(predicted_iou))
    keep = mask_nms(seg, scores, **kwargs)
    return torch.sort(keep.cpu().long()).values


def mask2segmap_device(seg: torch.Tensor, stability_score) -> Tuple[torch.Tensor, torch.Tensor]:
    """`mask2segmap` (segment_utils.py:12-27, sort=True) on the GPU: masks ordered by descending stability (stable),
    seg_map i32 [H, W] = first mask covering the pixel (-1 = none), binary_maps bool [N, H, W] in that order."""
    n, h, w = seg.shape
    order = np.argsort(-np.asarray(stability_score, np.float32), kind="stable")
    m = seg if seg.dtype == torch.uint8 else seg.view(torch.uint8)
    if (h * w) % 16 == 0 and m.is_contiguous():
        ordered = L.gather_rows(m, order.tolist())
    else:
        ordered = m.index_select(0, torch.from_numpy(order).to(m.device))
    seg_map = torch.empty((h, w), dtype=torch.int32, device=m.device)
    L.check(L.load().ovo_paint_segmap(L.ptr(ordered), n, h * w, L.ptr(seg_map), L.stream()))
    return seg_map, ordered.view(torch.bool)


def mask2segmap(masks: List[Dict[str, Any]], image: np.ndarray, sort: bool = True) -> Tuple[np.ndarray, np.ndarray]:
    """Reference: segment_utils.py:12-27.  i32[H, W] seg map (-1 = none) + bool[N, H, W]; most stable mask wins."""
    if sort:
        masks = heapq.nlargest(len(masks), masks, key=lambda m: m["stability_score"])
    binary_maps = np.stack([m["segmentation"] for m in masks])
    seg_map = np.full(image.shape[:2], -1, dtype=np.int32)
    for i, mk in enumerate(binary_maps):
        seg_map[mk & (seg_map < 0) if sort else mk] = i
    return seg_map, binary_maps


def batched_mask_to_box(masks: torch.Tensor) -> torch.Tensor:
    """Reference: segment_utils.py:43-94.  [N, H, W] bool -> i64[N, 4] XYXY, zeros for an empty mask."""
    if masks.numel() == 0:
        return torch.zeros(*masks.shape[:-2], 4, device=masks.device)
    h, w = masks.shape[-2:]
    m = masks.reshape(-1, h, w).bool()
    rows, cols = m.any(-1), m.any(-2)
    ys = torch.arange(h, device=m.device)
    xs = torch.arange(w, device=m.device)
    bottom = (rows * ys).amax(-1)
    top = torch.where(rows, ys, h).amin(-1)
    right = (cols * xs).amax(-1)
    left = torch.where(cols, xs, w).amin(-1)
    box = torch.stack([left, top, right, bottom], dim=-1)
    box = box * ~((right < left) | (bottom < top)).unsqueeze(-1)
    return box.reshape(*masks.shape[:-2], 4)


def batched_box_xyxy_to_xywh(box_xyxy: torch.Tensor) -> torch.Tensor:
    """Reference: segment_utils.py:88-94 (in place, like the reference): w = x2 - x1, h = y2 - y1 over INCLUSIVE edges."""
    box_xyxy[:, 2] = box_xyxy[:, 2] - box_xyxy[:, 0]
    box_xyxy[:, 3] = box_xyxy[:, 3] - box_xyxy[:, 1]
    return box_xyxy


def increase_bbox_by_margin(bbox, margin: int):
    """Reference: segment_utils.py:152-172."""
    x, y, w, h = (int(v) for v in bbox)
    x, y, w, h = x - margin, y - margin, w + 2 * margin, h + 2 * margin
    if x < 0:
        w, x = w + x, 0
    if y < 0:
        h, y = h + y, 0
    return x, y, w, h


def mask_boxes_xywh(binary_map: torch.Tensor) -> torch.Tensor:
    """batched_mask_to_box followed by batched_box_xyxy_to_xywh in ONE launch -> i32 [N, 4] on the GPU."""
    m = binary_map if binary_map.dtype == torch.uint8 else binary_map.view(torch.uint8) if binary_map.dtype == torch.bool else binary_map.to(torch.uint8)
    m = L.dev(m.contiguous(), torch.uint8, "binary_map")
    n, h, w = m.shape
    boxes = torch.empty((n, 4), dtype=torch.int32, device=m.device)
    L.check(L.load().ovo_mask_boxes(L.ptr(m), n, h, w, L.ptr(boxes), L.stream()))
    return boxes


def segmap2segimg(binary_map: torch.Tensor, image: torch.Tensor, also_bbox: bool, bbox_margin: int = 50, out_l: int = 224) -> torch.Tensor:
    """Reference: segment_utils.py:29-41 (+ :118-150).  binary_map bool [N, H, W], image [3, H, W] (u8 or float, 0..255)
    -> f32 [N, 3 | 6, out_l, out_l]: the masked crop (zero background; zero-padded to a square when there is no bbox part)
    and, with `also_bbox`, the box crop grown by `bbox_margin`, each resized like torchvision's F.resize (bilinear,
    antialias).  A uint8 image gives rounded values, as F.resize does for uint8 tensors.  All masks in one launch."""
    n = int(binary_map.shape[0])
    parts = 6 if also_bbox else 3
    img = L.dev(image.contiguous(), image.dtype if image.dtype == torch.uint8 else torch.float32, "image") \
        if image.dtype in (torch.uint8, torch.float32) else L.dev(image.float().contiguous(), torch.float32, "image")
    out = torch.empty((n, parts, out_l, out_l), dtype=torch.float32, device=img.device)
    if n == 0:
        return out
    m = binary_map.view(torch.uint8) if binary_map.dtype == torch.bool else binary_map.to(torch.uint8)
    m = L.dev(m.contiguous(), torch.uint8, "binary_map")
    _, h, w = img.shape
    if tuple(m.shape[1:]) != (h, w):
        raise L.OvoHipError(f"mask size {tuple(m.shape[1:])} != image size {(h, w)}")
    boxes = mask_boxes_xywh(m)
    L.check(L.load().ovo_mask_crops(L.ptr(img), L.DTYPE_CODE[img.dtype], h, w, L.ptr(m), L.ptr(boxes), n, int(also_bbox), int(bbox_margin),
                                    int(out_l), int(img.dtype == torch.uint8), L.ptr(out), L.stream()))
    return out
