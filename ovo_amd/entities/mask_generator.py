"""2D mask provider: SAM2 image encoder on MI355X + mask NMS / seg-map painting.

Mirror of the reference's `ovo/entities/mask_generator.py:MaskGenerator` (same constructor, `get_masks`,
`segment`, `precompute`, `.npy` cache names -- SURVEY.md §8b).

Scope (SURVEY.md §8 a10 + f1): the SAM2 image encoder (`ovo_amd.encoders.hiera.HipHiera`), the prompt encoder + mask
decoder (`ovo_amd.encoders.sam_decoder.HipSamDecoder`) and the automatic mask generator
(`ovo_amd.entities.sam_amg.HipSam2AutomaticMaskGenerator`) all run on the GPU; the masks go through the same NMS and
seg-map painting as the reference (mask_generator.py:118-119) without leaving HBM.  Masks can still come from the
reference's precomputed-mask seam (mask_generator.py:94-95,170-195: `sam.precomputed: True` + `.npy` files) or from an
injected `mask_source(image, image_embeddings) -> list[dict]` of SAM-style mask dicts.
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
        """Reference: mask_generator.py:39-53 + segment_utils.py:262-308 (loads SAM2 and builds its automatic mask
        generator with points_per_side / pred_iou_thresh (sic: `nms_iou_th`) / stability_score_thresh from the config).
        Here: Hiera image encoder + mask decoder + generator, all on the GPU (random-init unless `sam_ckpt_state` /
        a state dict is supplied -- there are no checkpoints offline)."""
        from ..encoders.hiera import SPECS, HipHiera
        from ..encoders.sam_decoder import SPECS as DSPECS, HipSamDecoder
        from .sam_amg import HipSam2AutomaticMaskGenerator
        enc = config.get("sam_encoder", "hiera_l")
        if enc not in SPECS:
            raise NotImplementedError(f"sam_encoder {enc}: supported here: {sorted(SPECS)} (SAM1 ViT encoders are not built)")
        state = config.get("sam_state")                           # optional sam2-style state dict (trunk.*, neck.*, sam_mask_decoder.*, ...)
        if self.image_encoder is None:
            self.image_encoder = HipHiera(SPECS[enc], state, device=self.device, seed=config.get("seed", 0))
        dspec = DSPECS[config.get("sam_decoder", "sam2")]
        decoder = HipSamDecoder(dspec, state, device=self.device, seed=config.get("seed", 0))
        self.mask_generator = HipSam2AutomaticMaskGenerator(
            self.image_encoder, decoder, points_per_side=config.get("points_per_side", 32),
            pred_iou_thresh=config.get("nms_iou_th", 0.8), stability_score_thresh=config.get("stability_score_th", 0.95),
            min_mask_region_area=config.get("min_mask_region_area", 0), use_m2m=config.get("use_m2m", False))

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
        dev = "cuda" if self.device == "cpu" else self.device
        if self.precomputed:
            seg_map, binary_maps = self._load_masks(frame_id)
        elif self.mask_source is None:                            # native generator: the masks never leave the GPU
            return self.segment_device(image)
        else:
            seg_map, binary_maps = self.segment(image)
        return torch.from_numpy(seg_map).to(dev), torch.from_numpy(binary_maps).to(dev)

    @torch.no_grad()
    def encode(self, image: np.ndarray):
        """SAM2 image-encoder forward for one u8[H,W,3] frame -> dict of feature maps (kept in `last_embeddings`)."""
        if self.image_encoder is None:
            self.load_mask_generator(self.config)
        self.last_embeddings = self.image_encoder.encode_frame(image)
        return self.last_embeddings

    @torch.no_grad()
    def segment_device(self, image) -> Tuple[torch.Tensor, torch.Tensor]:
        """`segment` with every stage on the GPU: generator -> masks_update (NMS) -> mask2segmap."""
        if self.mask_generator is None:
            self.load_mask_generator(self.config)
        r = self.mask_generator.generate_device(image)
        self.last_embeddings = self.mask_generator.last_embeddings
        masks = r["masks"]
        dev = masks.device
        if masks.shape[0] == 0:
            return torch.empty(0, device=dev), torch.empty(0, device=dev)
        keep = segment_utils.masks_update_device(masks, r["predicted_iou"], r["stability_score"], iou_thr=self.nms_iou_th,
                                                 score_thr=self.nms_score_th, inner_thr=self.nms_inner_th)
        kept = masks.index_select(0, keep.to(dev))
        return segment_utils.mask2segmap_device(kept, r["stability_score"][keep.numpy()])

    @torch.no_grad()
    def segment(self, image: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        """Reference: mask_generator.py:102-120."""
        if self.mask_source is None:
            seg_map, binary_maps = self.segment_device(image)
            return seg_map.cpu().numpy(), binary_maps.cpu().numpy()
        emb = self.encode(image)
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
