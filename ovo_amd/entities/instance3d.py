"""Per-3D-instance bookkeeping; descriptors live in a device-resident DescriptorBank.

Mirror of the reference's `ovo/entities/instance3d.py:Instance3D` (same public attributes and methods,
same top-k keyframe semantics and `to_update` rules, instance3d.py:77-155) with the heavy part moved:
the fused descriptor is a row of `DescriptorBank.table` on the GPU and multi-view fusion is the
`ovo_fuse_views` HIP kernel instead of per-instance CPU torch (instance3d.py:9-21,157-189).
An instance created without a bank (checkpoint restore, run_eval.py:19-28) holds a plain tensor.
"""
from __future__ import annotations

import bisect
import heapq
from typing import Any, Dict, List, Optional, Sequence

import numpy as np
import torch

from .descriptor_bank import FUSION_MODES, DescriptorBank


class Instance3D:
    n_top_kf: int = 0                 # 0 / negative: use every keyframe (instance3d.py:47)
    mv_fusion: str = "l1_medoid"      # name of the fusion rule (the reference stores the function)

    def __init__(self, id: int, kf_id: Optional[int] = None, points_ids: Optional[Sequence[int]] = None,
                 mask_area: int = 0, bank: Optional[DescriptorBank] = None):
        self.id = id
        self.kfs_ids: List[int] = []
        self.points_ids: List[Any] = []
        self._top_kf: List[tuple] = []         # min-heap of (area, kf_id)            } one set of entries in three shapes: the heap the
        self._top_area: Dict[int, Any] = {}    # kf_id -> area of its heap entry       } reference keeps (and exports), an index for the
        self._top_sorted: List[tuple] = []     # the entries in ascending order        } per-mask look-ups, the fusion order ready-made
        self.to_update = False
        # running-sum fusion (avg_pooling, OVO._planned_updates): the keyframes already in the sum, those pushed since the last plan, and whether
        # the sum has to be rebuilt (nothing summed yet, a summed view left the heap, or the descriptor was fused the full way in between)
        self._inc_kfs: set = set()
        self._pending_kfs: List[int] = []
        self._needs_full = True
        self._bank = bank
        self._own_feature: Optional[torch.Tensor] = None
        self._own_feature_kf = None
        if kf_id is not None:
            self.update(points_ids if points_ids is not None else [], kf_id, mask_area)

    # ------------------------------------------------------------------ class-level configuration
    @staticmethod
    def set_fusion(fusion: str, ckpt=None) -> None:
        if fusion == "camfusion":
            raise NotImplementedError("camfusion: the reference's loader is itself unimplemented (clip_utils.py:114-115)")
        if fusion not in FUSION_MODES:
            raise NotImplementedError(fusion)
        Instance3D.mv_fusion = fusion

    # ------------------------------------------------------------------ descriptor access
    @property
    def clip_feature(self) -> Optional[torch.Tensor]:
        if self._own_feature is not None:
            return self._own_feature
        if self._bank is not None and self._bank.has_feature(self.id):
            return self._bank.feature(self.id)
        return None

    @clip_feature.setter
    def clip_feature(self, value) -> None:
        if value is not None and self._bank is not None:
            self._bank.set_feature(self.id, torch.as_tensor(value))
            self._own_feature = None
        else:
            self._own_feature = value

    @property
    def clip_feature_kf(self):
        if self._own_feature is None and self._bank is not None and self._bank.has_feature(self.id):
            return self._bank.medoid_of.get(self._bank.slot_of[self.id])
        return self._own_feature_kf

    @clip_feature_kf.setter
    def clip_feature_kf(self, value) -> None:
        self._own_feature_kf = value

    # ------------------------------------------------------------------ observations (instance3d.py:77-155)
    def update(self, points_ids: Sequence[int], kf_id: int, area: int) -> None:
        self.add_keyframes(kf_id)
        self.add_points_ids(points_ids)
        self.add_top_kf(kf_id, area)

    def observe(self, kf_id: int, area: int) -> None:
        """`update([], kf_id, area)` for the tracking loop (one call per matched mask and keyframe, ovo.py:262-266): the same state changes as
        add_keyframes + add_top_kf with the usual case -- a keyframe this instance has not been seen in, a heap that is not full -- in line."""
        k = self.kfs_ids
        if not k or (k[-1] != kf_id and kf_id not in k):
            k.append(kf_id)
            if self.n_top_kf <= 0:
                self._pending_kfs.append(kf_id)
        top = self._top_area
        if kf_id not in top and len(self._top_kf) < self.n_top_kf:
            entry = (area, kf_id)
            heapq.heappush(self._top_kf, entry)
            top[kf_id] = area
            bisect.insort(self._top_sorted, entry)
            self._pending_kfs.append(kf_id)
            self.to_update = True
        else:
            self.add_top_kf(kf_id, area)

    def add_points_ids(self, points_ids) -> None:
        self.points_ids.extend(points_ids)

    def add_keyframes(self, kf_id: int) -> None:
        if not self.kfs_ids or (self.kfs_ids[-1] != kf_id and kf_id not in self.kfs_ids):      # (usually the keyframe just added, or a new one)
            self.kfs_ids.append(kf_id)
            if self.n_top_kf <= 0:
                self._pending_kfs.append(kf_id)

    @property
    def top_kf(self) -> List[tuple]:
        return self._top_kf

    @top_kf.setter
    def top_kf(self, entries) -> None:
        self._needs_full = True
        self._top_kf = list(entries)
        self._top_area = {kf: area for area, kf in self._top_kf}
        self._top_sorted = sorted(self._top_kf)

    def idx_in_top_kf(self, kf_id: int) -> int:
        if kf_id not in self._top_area:
            return -1
        for pos, entry in enumerate(self._top_kf):
            if entry[1] == kf_id:
                return pos
        return -1

    def is_top_kf(self, kf_id: int) -> bool:
        return kf_id in self._top_area

    def add_top_kf(self, kf_id: int, area: int) -> None:
        old = self._top_area.get(kf_id)
        if old is not None:
            if area > old:                           # same keyframe seen with a larger (fused) mask
                self._top_kf[self._top_kf.index((old, kf_id))] = (area, kf_id)
                heapq.heapify(self._top_kf)
                del self._top_sorted[bisect.bisect_left(self._top_sorted, (old, kf_id))]
                bisect.insort(self._top_sorted, (area, kf_id))
                self._top_area[kf_id] = area
                self.to_update = True
            return
        self._add_top_kf(kf_id, area)

    def _add_top_kf(self, kf_id: int, area: int) -> None:
        if len(self._top_kf) < self.n_top_kf:
            heapq.heappush(self._top_kf, (area, kf_id))
            self._top_area[kf_id] = area
            bisect.insort(self._top_sorted, (area, kf_id))
            self._pending_kfs.append(kf_id)
            self.to_update = True
            return
        evicted = heapq.heappushpop(self._top_kf, (area, kf_id))
        if self.n_top_kf <= 0 or evicted[1] != kf_id:
            self.to_update = True
        if evicted[1] != kf_id:                      # the new entry stayed, the smallest one left
            del self._top_area[evicted[1]]
            del self._top_sorted[bisect.bisect_left(self._top_sorted, evicted)]
            self._top_area[kf_id] = area
            bisect.insort(self._top_sorted, (area, kf_id))
            self._pending_kfs.append(kf_id)
            if evicted[1] in self._inc_kfs:
                self._needs_full = True

    # ------------------------------------------------------------------ fusion
    def fusion_views(self) -> List[int]:
        """Keyframes whose descriptors are fused, in the reference's stacking order (instance3d.py:170-178:
        `heapq.nlargest(n_top_kf, top_kf)` = the heap's entries in descending (area, kf) order -- it never holds more than n_top_kf)."""
        if self.n_top_kf > 0:
            return [kf for _, kf in reversed(self._top_sorted)]
        return list(self.kfs_ids)

    def update_clip(self, keyframes_clips: Dict[int, Any], force_update: bool = False) -> None:
        """Reference: instance3d.py:157-189.  `keyframes_clips[kf]` is a KeyframeView (or a dict of tensors)."""
        if not (self.to_update or force_update):
            return
        rows, loose = [], []
        for kf in self.fusion_views():
            view = keyframes_clips.get(kf)
            if view is None:
                continue
            if hasattr(view, "row"):
                rows.append(view.row(self.id))
            else:
                loose.append(view[self.id])
        if loose:                                   # plain tensors (restored checkpoints): adopt them into the bank
            bank = self._require_bank(loose[0])
            rows = rows + bank.append(torch.stack([t.reshape(-1) for t in loose]))
        if not rows:
            return
        self._own_feature = None
        self._require_bank(None).fuse([(self.id, rows)], Instance3D.mv_fusion)
        self._needs_full = True                      # (fused the full way: a later running-sum update starts over)
        self.to_update = False

    def _require_bank(self, like) -> DescriptorBank:
        if self._bank is None:
            if like is None:
                raise RuntimeError("Instance3D has no DescriptorBank")
            self._bank = DescriptorBank(like.numel(), like.device if like.is_cuda else "cuda")
        return self._bank

    # ------------------------------------------------------------------ checkpoint (instance3d.py:191-226)
    def export(self, debug_info: bool = False) -> Dict[str, Any]:
        feat = self.clip_feature
        out = {f"ins3d_{self.id}_clip_feature": None if feat is None else feat.detach().cpu().clone(),
               f"ins3d_{self.id}_clip_feature_kf": self.clip_feature_kf}
        if debug_info:
            out[f"ins3d_{self.id}_keyframes_ids"] = np.array(self.kfs_ids)
            out[f"ins3d_{self.id}_points_ids"] = np.array(self.points_ids)
            out[f"ins3d_{self.id}_top_kfs"] = np.array(self.top_kf)
        return out

    def restore(self, obj_dict: Dict[str, Any], debug_info: bool) -> None:
        self._own_feature = obj_dict[f"ins3d_{self.id}_clip_feature"]
        self._own_feature_kf = obj_dict.get(f"ins3d_{self.id}_clip_feature_kf")
        self.to_update = self._own_feature is None
        if debug_info:
            self.kfs_ids = obj_dict[f"ins3d_{self.id}_keyframes_ids"].tolist()
            self.points_ids = obj_dict[f"ins3d_{self.id}_points_ids"].tolist()
            top = obj_dict.get(f"ins3d_{self.id}_top_kfs")
            if top is not None:
                self.top_kf = [(area, kf) for area, kf in top]

    def purge_points_ids(self, purge_ids: Sequence[int]) -> None:
        self.points_ids = [p for p in self.points_ids if p not in purge_ids]
