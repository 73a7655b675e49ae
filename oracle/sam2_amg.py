"""Oracle: CPU restatement of the post-processing of SAM2's automatic mask generator (single crop).

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED: the reference calls
`sam2.automatic_mask_generator.SAM2AutomaticMaskGenerator.generate` (mask_generator.py:113, built in
segment_utils.py:291-308); neither `sam2` nor `torchvision` (its box NMS) is installed here.  The steps below restate
the published pipeline with torch / numpy primitives:
    postprocess_masks      F.interpolate(low_res, (H, W), mode="bilinear", align_corners=False)
    pred_iou filter        iou_preds > pred_iou_thresh
    stability score        #(logit > thr + offset) / #(logit > thr - offset)  >= stability_score_thresh
    binarise, boxes        logit > thr ; batched_mask_to_box (inclusive XYXY, zeros when empty)
    box NMS                torchvision.ops.nms semantics: greedy by descending predicted IoU, suppress IoU > box_nms_thresh
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from .features import masks_to_boxes


def box_nms(boxes: np.ndarray, scores: np.ndarray, thr: float) -> np.ndarray:
    order = np.argsort(-scores, kind="stable")
    keep, dead = [], set()
    for a, i in enumerate(order):
        if i in dead:
            continue
        keep.append(i)
        for j in order[a + 1:]:
            if j in dead:
                continue
            x0, y0 = max(boxes[i, 0], boxes[j, 0]), max(boxes[i, 1], boxes[j, 1])
            x1, y1 = min(boxes[i, 2], boxes[j, 2]), min(boxes[i, 3], boxes[j, 3])
            inter = np.float32(max(x1 - x0, 0)) * np.float32(max(y1 - y0, 0))
            ai = np.float32(boxes[i, 2] - boxes[i, 0]) * np.float32(boxes[i, 3] - boxes[i, 1])
            aj = np.float32(boxes[j, 2] - boxes[j, 0]) * np.float32(boxes[j, 3] - boxes[j, 1])
            union = ai + aj - inter
            if union > 0 and inter / union > np.float32(thr):
                dead.add(j)
    return np.asarray(keep, dtype=np.int64)


def amg_postprocess(logits: np.ndarray, iou: np.ndarray, H: int, W: int, pred_iou_thresh: float = 0.8, stability_score_thresh: float = 0.95,
                    offset: float = 1.0, thr: float = 0.0, box_nms_thresh: float = 0.7) -> Dict[str, np.ndarray]:
    """logits f32 [P, m, h, w], iou f32 [P, m] -> kept masks (bool [n, H, W]) in descending predicted-IoU order with their
    scores, boxes and flat candidate index, plus the per-candidate upsampled statistics (for tolerance-aware comparisons)."""
    P, m, h, w = logits.shape
    up = F.interpolate(torch.from_numpy(logits).reshape(1, P * m, h, w), (H, W), mode="bilinear", align_corners=False)[0]
    hi = (up > thr + offset).flatten(1).sum(1).numpy().astype(np.int64)
    lo = (up > thr - offset).flatten(1).sum(1).numpy().astype(np.int64)
    with np.errstate(divide="ignore", invalid="ignore"):
        stab = (hi.astype(np.float32) / lo.astype(np.float32)).astype(np.float32)
    flat_iou = iou.reshape(-1).astype(np.float32)
    cand = np.nonzero((flat_iou > np.float32(pred_iou_thresh)) & (stab >= np.float32(stability_score_thresh)))[0]
    masks = (up[cand] > thr).numpy()
    boxes = masks_to_boxes(masks).astype(np.float32)
    keep = box_nms(boxes, flat_iou[cand], box_nms_thresh)
    return {"masks": masks[keep], "predicted_iou": flat_iou[cand][keep], "stability_score": stab[cand][keep], "boxes_xyxy": boxes[keep],
            "index": cand[keep], "hi": hi, "lo": lo, "stab_all": stab}
