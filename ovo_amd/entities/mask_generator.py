"""2D mask provider: SAM2 image encoder on MI355X + mask NMS / seg-map painting.

Mirror of the reference's `ovo/entities/mask_generator.py:MaskGenerator` (same constructor, `get_masks`,
`segment`, `precompute`, `.npy` cache names -- SURVEY.md §8b).

Scope: the SAM2 *image encoder* (Hiera trunk + FPN neck, where the FLOPs are) runs
on the GPU through `ovo_amd.encoders.hiera.HipHiera`.  The prompt encoder, mask decoder and the
automatic-mask-generator post-processing are the next row (f1); until they exist the masks themselves come
from the reference's own precomputed-mask seam (mask_generator.py:94-95,170-195: `sam.precomputed: True` +
`.npy` files) or from an injected `mask_source(image, image_embeddings) -> list[dict]` producing SAM-style
mask dicts (`segmentation`, `predicted_iou`, `stability_score`), which then go through the same NMS and
seg-map painting as the reference (mask_generator.py:118-119).
"""
from __future__ import annotations

import os
from typing import Any, Callable, Dict, Optional, Tuple

import numpy as np
import torch

from ..utils import segment_utils


class MaskGenerator:
    def __init__(self, config: Dict[str, Any], scene_name: Optional[str] = None, device="cuda",
                 image_encoder=None, mask_source: Optional[Callable] = None) -> None:
        self.precomputed = config.get("precomputed", False)
        self.config = config
        if scene_name:
            self.masks_path = os.path.join(config.get("masks_base_path", ""), scene_name)
        else:
            assert not config.get("precompute", False), "To precompute masks or use precomputed masks \"scene_name\" is required!"
            self.masks_path = ""
        self.nms_iou_th = config.get("nms_iou_th", 0.8)
        self.nms_score_th = config.get("nms_score_th", 0.7)
        self.nms_inner_th = config.get("nms_inner_th", 0.5)
        self.multi_crop = config.get("multi_crop", False)
        self.device = device
        self.image_encoder = image_encoder
        self.mask_source = mask_source
        self.mask_generator = None                      # the reference's attribute; None = use cached masks
        self.last_embeddings = None
        if (self.precomputed or config.get("precompute", False)) and os.path.isdir(self.masks_path):
            print(" {} path already exists, skipping masks precompute! To recompute masks delete old masks!".format(self.masks_path))
        elif image_encoder is None and not self.precomputed and mask_source is None:
            self.load_mask_generator(config)

    def load_mask_generator(self, config: Dict[str, Any]) -> None:
        """Reference: mask_generator.py:39-53 (loads SAM/SAM2 and warms it up).  Here: the Hiera image encoder."""
        from ..encoders.hiera import SPECS, HipHiera
        enc = config.get("sam_encoder", "hiera_l")
        if enc not in SPECS:
            raise NotImplementedError(f"sam_encoder {enc}: supported here: {sorted(SPECS)} (SAM1 ViT encoders are not built)")
        self.image_encoder = HipHiera(SPECS[enc], None, device=self.device, seed=config.get("seed", 0))

    def to(self, device: str) -> None:
        self.device = device

    def cpu(self) -> None:
        self.device = "cpu"           # weights stay on the GPU; there is no CPU execution path

    def cuda(self) -> None:
        self.device = "cuda"

    # ------------------------------------------------------------------ API used by OVO._get_masks
    def get_masks(self, image: np.ndarray, frame_id: Optional[int] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """Reference: mask_generator.py:81-99 -> (seg_map i32[H,W], binary_maps bool[N,H,W]) on the device;
        two empty tensors when nothing was segmented."""
        if self.precomputed:
            seg_map, binary_maps = self._load_masks(frame_id)
        else:
            seg_map, binary_maps = self.segment(image)
        dev = "cuda" if self.device == "cpu" else self.device
        return torch.from_numpy(seg_map).to(dev), torch.from_numpy(binary_maps).to(dev)

    @torch.no_grad()
    def encode(self, image: np.ndarray):
        """SAM2 image-encoder forward for one u8[H,W,3] frame -> dict of feature maps (kept in `last_embeddings`)."""
        if self.image_encoder is None:
            self.load_mask_generator(self.config)
        self.last_embeddings = self.image_encoder.encode_frame(image)
        return self.last_embeddings

    @torch.no_grad()
    def segment(self, image: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """Reference: mask_generator.py:102-120."""
        emb = self.encode(image)
        if self.mask_source is None:
            raise NotImplementedError("SAM2 prompt encoder / mask decoder / automatic mask generator are the next row "
                                      "(SURVEY.md §8 f1): use sam.precomputed masks or inject mask_source")
        masks = self.mask_source(image, emb)
        if len(masks) == 0:
            return np.array([]), np.array([])
        kept, = segment_utils.masks_update(masks, iou_thr=self.nms_iou_th, score_thr=self.nms_score_th,
                                           inner_thr=self.nms_inner_th)
        return segment_utils.mask2segmap(kept, image)

    # ------------------------------------------------------------------ .npy mask cache (mask_generator.py:122-195)
    def _paths(self, frame_id: int) -> Tuple[str, str]:
        return (os.path.join(self.masks_path, f"{frame_id:04d}_seg_map_default.npy"),
                os.path.join(self.masks_path, f"{frame_id:04d}_bmap_default.npy"))

    def precompute(self, dataset, segment_every: int) -> None:
        print("Precomputing segmentation masks.")
        os.makedirs(self.masks_path, exist_ok=True)
        for frame_id in range(0, len(dataset), segment_every):
            seg_path, bmap_path = self._paths(frame_id)
            if os.path.exists(seg_path) and os.path.exists(bmap_path):
                print(f"Frame {frame_id} already compute. Skipping ...")
                continue
            seg_map, binary_maps = self.segment(dataset[frame_id][1])
            self._save_masks(seg_map, binary_maps, frame_id)
        self.precomputed = True

    def _save_masks(self, seg_map: np.ndarray, binary_maps: np.ndarray, frame_id: int) -> None:
        seg_path, bmap_path = self._paths(frame_id)
        np.save(seg_path[:-4], seg_map)
        np.save(bmap_path[:-4], binary_maps)

    def _load_masks(self, frame_id: int) -> Tuple[np.ndarray, np.ndarray]:
        seg_path, bmap_path = self._paths(frame_id)
        if not os.path.exists(seg_path):
            print(f"No precomputed mask for frame {frame_id}")
            return np.array([]), np.array([])
        seg_map = np.load(seg_path)
        if os.path.exists(bmap_path):
            return seg_map, np.load(bmap_path)
        ids = np.arange(seg_map.max() + 1)
        return seg_map, seg_map[None] == ids[:, None, None]
