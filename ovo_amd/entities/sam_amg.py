"""SAM2 automatic mask generator on MI355X.

The reference builds `sam2.automatic_mask_generator.SAM2AutomaticMaskGenerator(model, points_per_side, pred_iou_thresh,
stability_score_thresh, min_mask_region_area, use_m2m)` (segment_utils.py:291-308) and calls `.generate(image)`
(mask_generator.py:113).  That package is not vendored; this class restates its published single-crop pipeline
(crop_n_layers = 0, multimask_output = True, min_mask_region_area = 0, use_m2m = False -- the reference's settings):

    image encoder -> regular grid of foreground clicks -> mask decoder (3 masks + predicted IoU per click)
    -> keep predicted IoU > pred_iou_thresh -> stability score  #(logit > +offset) / #(logit > -offset) >= thresh
    -> binarise at 0 -> boxes -> box NMS (IoU > 0.7 suppressed, by predicted IoU) -> SAM-style records

MI355X design: all clicks of the grid go through the decoder in one batch (`HipSamDecoder`); the filters run on the
256 x 256 logits through their H x W bilinear upsampling evaluated on the fly (`ovo_amg_mask_stats`: one pass, 7 integers
per candidate, one small D2H copy); only the masks that survive NMS are ever written at full resolution
(`ovo_amg_binarize`).  The reference materialises 3 x points full-resolution logit maps per batch of 64 clicks.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch

from .. import _lib as L


def box_nms(boxes: np.ndarray, scores: np.ndarray, iou_threshold: float) -> np.ndarray:
    """torchvision.ops.nms restated (greedy, descending score, suppress IoU > threshold; areas (x2-x1)*(y2-y1)).
    boxes f32 [n, 4] XYXY -> kept indices in descending-score order."""
    order = np.argsort(-scores, kind="stable")
    b = boxes[order].astype(np.float32)
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    alive = np.ones(len(order), bool)
    keep = []
    for i in range(len(order)):
        if not alive[i]:
            continue
        keep.append(order[i])
        w = np.clip(np.minimum(b[i, 2], b[i + 1:, 2]) - np.maximum(b[i, 0], b[i + 1:, 0]), 0, None)
        h = np.clip(np.minimum(b[i, 3], b[i + 1:, 3]) - np.maximum(b[i, 1], b[i + 1:, 1]), 0, None)
        inter = w * h
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = inter / (area[i] + area[i + 1:] - inter)
        alive[i + 1:] &= ~(iou > np.float32(iou_threshold))
    return np.asarray(keep, dtype=np.int64)


def remove_small_regions(mask: np.ndarray, area_thresh: float, mode: str):
    """sam2 / segment_anything `utils.amg.remove_small_regions` restated [upstream-knowledge, UNPINNED: neither package is available offline]:
    mode "holes" fills background components smaller than `area_thresh` that the mask encloses or touches, mode "islands" removes foreground
    components smaller than it (if ALL are smaller, the largest stays).  8-connectivity, like cv2.connectedComponentsWithStats(m, 8) there;
    scipy.ndimage.label gives the same components (component numbering differs, the result does not depend on it).  Returns (mask, changed)."""
    from scipy import ndimage
    assert mode in ("holes", "islands")
    correct_holes = mode == "holes"
    working = np.logical_xor(correct_holes, mask.astype(bool))
    regions, n = ndimage.label(working, structure=np.ones((3, 3), dtype=bool))
    sizes = np.bincount(regions.reshape(-1), minlength=n + 1)[1:]                  # label 0 = background of `working`
    small = [i + 1 for i, sz in enumerate(sizes) if sz < area_thresh]
    if len(small) == 0:
        return mask.astype(bool), False
    fill = [0] + small
    if not correct_holes:
        fill = [i for i in range(n + 1) if i not in fill]
        if len(fill) == 0:                                                         # every region is below the threshold: keep the largest
            fill = [int(np.argmax(sizes)) + 1]
    return np.isin(regions, fill), True


def postprocess_small_regions(masks: np.ndarray, boxes_xyxy: np.ndarray, min_area: int, nms_thresh: float):
    """`SAM2AutomaticMaskGenerator.postprocess_small_regions` restated [UNPINNED]: holes, then islands, smaller than `min_area` are removed from
    every kept mask; boxes are recomputed; a box NMS that prefers UNCHANGED masks (score 1) over changed ones (score 0) drops duplicates the
    clean-up created.  masks bool [n, H, W], boxes i32 [n, 4] -> (masks, boxes, keep indices into the input, changed flags of the kept)."""
    if len(masks) == 0:
        return masks, boxes_xyxy, np.zeros(0, np.int64), np.zeros(0, bool)
    new, scores = [], []
    for m in masks:
        m1, ch1 = remove_small_regions(m, min_area, "holes")
        m2, ch2 = remove_small_regions(m1, min_area, "islands")
        new.append(m2)
        scores.append(0.0 if (ch1 or ch2) else 1.0)
    new = np.stack(new)
    nb = np.zeros((len(new), 4), np.int32)                                         # batched_mask_to_box: [x0, y0, x1, y1] inclusive, zeros if empty
    for i, m in enumerate(new):
        ys, xs = np.nonzero(m)
        if len(ys):
            nb[i] = (xs.min(), ys.min(), xs.max(), ys.max())
    scores = np.asarray(scores, np.float32)
    keep = box_nms(nb.astype(np.float32), scores, nms_thresh)
    out_boxes = boxes_xyxy.copy()
    changed = scores == 0.0
    out_boxes[changed] = nb[changed]                                               # only recalculated for masks that changed (as upstream)
    return new[keep], out_boxes[keep], keep, changed[keep]


class HipSam2AutomaticMaskGenerator:
    def __init__(self, image_encoder, decoder, points_per_side: int = 32, pred_iou_thresh: float = 0.8,
                 stability_score_thresh: float = 0.95, stability_score_offset: float = 1.0, mask_threshold: float = 0.0,
                 box_nms_thresh: float = 0.7, min_mask_region_area: int = 0, use_m2m: bool = False, **unused):
        if use_m2m:
            raise NotImplementedError("use_m2m is not built; the reference runs SAM2 with False (segment_utils.py:300-303)")
        self.min_mask_region_area = int(min_mask_region_area)        # > 0: host clean-up of the kept masks (segment_utils.py:283,300: SAM1 100, SAM2 0)
        self.encoder, self.decoder = image_encoder, decoder
        self.points_per_side = points_per_side
        self.pred_iou_thresh, self.stability_score_thresh = pred_iou_thresh, stability_score_thresh
        self.stability_score_offset, self.mask_threshold, self.box_nms_thresh = stability_score_offset, mask_threshold, box_nms_thresh
        self.grid01 = None
        self.last_embeddings = None
        self._pinned = None

    def _set_grid(self):
        if self.grid01 is None:
            from ..encoders.sam_decoder import point_grid
            self.grid01 = point_grid(self.points_per_side)
            self.decoder.set_points(self.grid01 * self.decoder.spec.image_size)

    @torch.no_grad()
    def generate_launch(self, image, embeddings: Optional[Dict[str, Any]] = None) -> Dict[str, Any]:
        """Enqueue encoder -> decoder -> candidate statistics on the current stream and start the (small) device->host copy of
        the statistics into pinned memory.  Nothing waits; `generate_finish` does.  Lets the caller overlap the generator with
        other streams' work (the ViT forward, back-projection) instead of parking the host in a sync.
        `embeddings`: the image encoder's output for this frame when it was computed elsewhere (one batched Hiera forward for
        several frames: dict(image_embed [1, s, s, C], high_res_feats (f0 [1, 4s, 4s, C/8], f1 [1, 2s, 2s, C/4])))."""
        self._set_grid()
        lib = L.load()
        if isinstance(image, np.ndarray):
            H, W = image.shape[:2]
        else:
            H, W = (image.shape[0], image.shape[1]) if image.shape[-1] == 3 else (image.shape[1], image.shape[2])
            if image.shape[-1] == 3:
                image = image.permute(2, 0, 1).contiguous()
        emb = embeddings if embeddings is not None else self.encoder.encode_frame(image)
        self.last_embeddings = emb
        f0, f1 = emb["high_res_feats"]
        logits, iou = self.decoder.forward(emb["image_embed"][0], f1[0], f0[0], multimask=True)       # [P, 3, h, w], [P, 3]
        return self.stats_launch(logits, iou, H, W)

    @torch.no_grad()
    def stats_launch(self, logits: torch.Tensor, iou: torch.Tensor, H: int, W: int) -> Dict[str, Any]:
        """The generator's post-processing from given mask logits f32 [P, m, h, w] and predicted IoUs f32 [P, m] (device tensors): the
        candidate statistics + their copy to pinned memory; `generate_finish` filters.  (`generate_launch` ends here; tools/check_upstream.py
        and the tests feed it logits that upstream's own post-processing has seen.)"""
        lib = L.load()
        self.last_logits, self.last_iou = logits, iou                # kept for inspection / parity tests
        P, nm, h, w = logits.shape
        n = P * nm
        stats = torch.empty((n, 7), dtype=torch.int32, device=logits.device)
        L.check(lib.ovo_amg_mask_stats(L.ptr(logits), n, h, w, H, W, float(self.mask_threshold), float(self.stability_score_offset),
                                       L.ptr(stats), L.stream()))
        if self._pinned is None or self._pinned[0].shape[0] != n:
            self._pinned = (torch.empty((n, 7), dtype=torch.int32).pin_memory(), torch.empty(n, dtype=torch.float32).pin_memory())
        self._pinned[0].copy_(stats, non_blocking=True)
        self._pinned[1].copy_(iou.reshape(-1), non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        return {"logits": logits, "iou": iou, "stats": stats, "done": done, "HW": (H, W), "stream": torch.cuda.current_stream()}

    @torch.no_grad()
    def generate_finish(self, h: Dict[str, Any]) -> Dict[str, Any]:
        """Wait for the statistics, filter + box-NMS on the host (a few hundred candidates), binarise the survivors."""
        lib = L.load()
        h["done"].synchronize()                                      # the one sync of the generator
        logits = h["logits"]
        P, nm, lh, lw = logits.shape
        H, W = h["HW"]
        st, iou_h = self._pinned[0].numpy().copy(), self._pinned[1].numpy().copy()
        with np.errstate(divide="ignore", invalid="ignore"):
            stab = (st[:, 0].astype(np.float32) / st[:, 1].astype(np.float32)).astype(np.float32)
        cand = np.nonzero((iou_h > np.float32(self.pred_iou_thresh)) & (stab >= np.float32(self.stability_score_thresh)))[0]
        boxes = st[cand, 3:7].copy()
        boxes[st[cand, 2] == 0] = 0                                 # empty mask -> [0, 0, 0, 0] like batched_mask_to_box
        keep = box_nms(boxes.astype(np.float32), iou_h[cand], self.box_nms_thresh)
        sel = cand[keep]
        with torch.cuda.stream(h["stream"]):
            masks = torch.empty((len(sel), H, W), dtype=torch.uint8, device=logits.device)
            if len(sel):
                d_sel = torch.from_numpy(sel.astype(np.int32)).to(logits.device, non_blocking=True)
                L.check(lib.ovo_amg_binarize(L.ptr(logits), L.ptr(d_sel), len(sel), lh, lw, H, W, float(self.mask_threshold), L.ptr(masks),
                                             L.stream()))
        out = {"masks": masks, "predicted_iou": iou_h[sel], "stability_score": stab[sel], "boxes_xyxy": boxes[keep],
               "area": st[sel, 2].copy(), "point_index": sel // nm, "index": sel}
        if self.min_mask_region_area > 0 and len(sel):
            # small disconnected regions and holes (generate(): `if self.min_mask_region_area > 0: postprocess_small_regions(...)`): a host pass over
            # the few dozen kept masks -- one device sync; not on the reference's SAM2 path (min_mask_region_area 0)
            with torch.cuda.stream(h["stream"]):
                host = masks.cpu().numpy().astype(bool)
                new, nboxes, keep2, changed = postprocess_small_regions(host, out["boxes_xyxy"], self.min_mask_region_area, self.box_nms_thresh)
                out = {k: (v[keep2] if k != "masks" else v) for k, v in out.items()}
                out["boxes_xyxy"] = nboxes
                out["area"] = np.where(changed, new.reshape(len(new), -1).sum(1), out["area"]).astype(out["area"].dtype)
                out["masks"] = torch.from_numpy(new.astype(np.uint8)).to(logits.device)
        return out

    def generate_device(self, image) -> Dict[str, Any]:
        """image u8 [H, W, 3] (numpy or device tensor) -> dict(masks u8 [n, H, W] on the GPU, predicted_iou f32 [n],
        stability_score f32 [n], boxes_xyxy i32 [n, 4], point_index i64 [n]), in descending predicted-IoU order."""
        return self.generate_finish(self.generate_launch(image))

    def generate(self, image) -> List[Dict[str, Any]]:
        """SAM-style records like `SAM2AutomaticMaskGenerator.generate` (segmentation as a numpy bool array)."""
        r = self.generate_device(image)
        seg = r["masks"].cpu().numpy().astype(bool)
        H, W = seg.shape[1:] if len(seg) else (0, 0)
        pts = (self.grid01.numpy() * np.array([W, H])) if len(seg) else None
        out = []
        for i in range(len(seg)):
            x0, y0, x1, y1 = (int(v) for v in r["boxes_xyxy"][i])
            out.append({"segmentation": seg[i], "area": int(r["area"][i]), "bbox": [x0, y0, x1 - x0, y1 - y0],
                        "predicted_iou": float(r["predicted_iou"][i]), "point_coords": [pts[r["point_index"][i]].tolist()],
                        "stability_score": float(r["stability_score"][i]), "crop_box": [0, 0, W, H]})
        return out
