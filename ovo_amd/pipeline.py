"""One keyframe of OVO's open-vocabulary feature path, end to end on one GPU.

This is the frame loop body of the reference's `OVOSemMap.run` (ovomapping.py:139-187) with every frame a semantic
keyframe (map_every = segment_every = 1, kf_queue_delay = 0), wired from this package's drop-in classes:

    slam.track_camera / slam.map          back-projection into the point map                     (a9)
    SAM2 image encoder                    Hiera + FPN on the 1024^2 resized frame                (a10)
    ovo.detect_and_track_objects          masks (precomputed-mask seam) -> cull / project / match / vote / assign (a1-a8)
    ovo.compute_semantic_info             ViT tokens -> region pooling -> multi-view fusion      (a12, a15-a17, a20)
    dense ("voxel") fusion                per-point scatter-accumulate of the mask descriptors   (BASELINE configs 3-4)
    query                                 instance table x texts (+argmax) and dense map x texts (a21-a22)

It is what bench.py times and what smoke() runs small; it adds no arithmetic of its own.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib as L
from . import parallel, synthetic as syn
from .encoders.hiera import SPECS as HIERA_SPECS, HipHiera
from .encoders.vit import SPECS as VIT_SPECS, HipViT
from .entities.clip_generator import CLIPGenerator
from .entities.ovo import OVO
from .slam.vanilla_mapper import VanillaMapper
from .utils import clip_utils


@dataclass
class Frame:
    index: int
    rgb: torch.Tensor          # u8 [H, W, 3]      colour frame (640x480 for ScanNet)
    rgb_lr: torch.Tensor       # u8 [h, w, 3]      colour at depth resolution (crop_edge applied)
    depth: torch.Tensor        # f32 [h, w]        metres, 0 = invalid
    c2w: np.ndarray            # f32 [4, 4]
    seg_map: torch.Tensor      # i32 [H, W]        -1 = no mask
    masks: torch.Tensor        # bool [N, H, W]
    ready: Optional[torch.cuda.Event] = None   # recorded after the frame's host->device upload (None: already resident)


class ResidentMasks:
    """The reference's precomputed-mask seam (mask_generator.py:94-95) with the masks already in HBM."""

    def __init__(self):
        self.frames: Dict[int, Frame] = {}
        self.precomputed = True

    def get_masks(self, image, frame_id):
        f = self.frames[frame_id]
        return f.seg_map, f.masks.clone()        # OVO fuses masks in place (ovo.py:303): hand out a copy


def synthetic_frames(n: int, device, scale: float = 1.0, n_masks_grid=(4, 6), n_blobs: int = 8, seed: int = 0, start: int = 0) -> List[Frame]:
    """Deterministic 640x480-style RGB-D frames with consistent geometry, resident on `device`."""
    h, w = syn.scannet_depth_hw(scale)
    e = int(round(syn.SCANNET["crop_edge"] * scale))
    H, W = h + 2 * e, w + 2 * e
    K = syn.scannet_intrinsics(scale)
    out = []
    for t in range(start, start + n):
        c2w = syn.pose(t % 24)                                     # the trajectory loops inside the room
        rgb = syn.render_rgb(H, W, seed + t)
        depth = syn.render_depth(c2w, K, h, w, seed + t)
        masks = syn.make_masks(H, W, grid=n_masks_grid, n_blobs=n_blobs, seed=seed + t)
        out.append(Frame(t, torch.from_numpy(rgb).to(device), torch.from_numpy(np.ascontiguousarray(rgb[e:H - e, e:W - e])).to(device),
                         torch.from_numpy(depth).to(device), c2w, torch.from_numpy(syn.masks_to_segmap(masks)).to(device),
                         torch.from_numpy(masks).to(device)))
    return out


class FramePipeline:
    def __init__(self, device="cuda", vit_card: str = "PE-Core-L14-336", sam_card: Optional[str] = "hiera_b+",
                 n_map: int = 1_000_000, n_text: int = 10, dense: bool = True, scale: float = 1.0, extra_capacity: int = 4_000_000,
                 seed: int = 0, depth_filter: bool = True, track_th: int = 100, sam_full: bool = False, points_per_side: int = 16,
                 encoder_batch: int = 1):
        self.device = torch.device(device)
        self.scale = scale
        self.crop_edge = int(round(syn.SCANNET["crop_edge"] * scale))
        K = torch.from_numpy(syn.scannet_intrinsics(scale)).to(self.device)
        self.slam = VanillaMapper({"device": str(self.device), "mapping": {"k_pooling": 3}}, K)
        if n_map > 0:
            pts = torch.from_numpy(syn.padded_map(n_map, frames=4, scale=scale, seed=seed))
            self.slam.set_map_dict({"xyz": pts, "obj_ids": torch.full((n_map, 1), -1, dtype=torch.int32),
                                    "ids": torch.arange(n_map, dtype=torch.int32)[:, None], "max_id": n_map,
                                    "color": torch.zeros((n_map, 3), dtype=torch.uint8)})
        self.slam._reserve(n_map + extra_capacity)
        self.masks = ResidentMasks()
        clip_cfg = {"embed_type": "TextRegion", "model_card": vit_card, "k_top_views": 10000, "fusion": "avg_pooling", "seed": seed}
        self.clip = CLIPGenerator(clip_cfg, device=str(self.device), encoder=HipViT(VIT_SPECS[vit_card], None, self.device, seed))
        cfg = {"match_distance_th": 0.05, "track_th": track_th, "depth_filter": depth_filter, "log": False, "kf_queue_delay": 0,
               "debug_info": False, "clip": clip_cfg, "sam": {"precomputed": True}}
        self.ovo = OVO(cfg, None, "synthetic", K, device=str(self.device), clip_generator=self.clip, mask_generator=self.masks)
        self.sam = HipHiera(HIERA_SPECS[sam_card], None, self.device, seed) if sam_card else None
        self.sam_out = None
        self.amg = None
        if sam_card and sam_full:                                  # SAM2 end to end (f1): decoder + automatic mask generator after the encoder
            from .encoders.sam_decoder import SPECS as DEC_SPECS, HipSamDecoder
            from .entities.sam_amg import HipSam2AutomaticMaskGenerator
            self.amg = HipSam2AutomaticMaskGenerator(self.sam, HipSamDecoder(DEC_SPECS["sam2"], None, self.device, seed), points_per_side=points_per_side)
        # encoder look-ahead: the two encoders of `encoder_batch` consecutive keyframes run as ONE batched forward each (step() is
        # handed the upcoming frames).  Nothing of the encoders depends on the map, and the reference itself computes a keyframe's
        # descriptors kf_queue_delay = 10 keyframes late (ovo.yaml:53), so this changes no result -- only the GEMM height.
        self.encoder_batch = max(1, int(encoder_batch)) if not sam_full else 1
        self._encoded: Dict[int, bool] = {}
        self.serial = False                                        # measurement only: both encoders on the caller's stream
        self._group_first: Dict[int, int] = {}
        self._sam_in = None
        self.prefetch = not os.environ.get("OVO_NO_PREFETCH")
        self.join_each_step = bool(os.environ.get("OVO_JOIN_EACH_STEP"))
        self.sam_stream = torch.cuda.Stream(device=self.device, priority=int(os.environ.get("OVO_SAM_PRIORITY", "0"))) if (sam_card and not os.environ.get("OVO_SAM_SAME_STREAM")) else None
        self.D = self.clip.clip_dim
        self.texts = torch.from_numpy(syn.unit_vectors(n_text, self.D, seed=seed + 7)).to(self.device)
        self.dense = dense
        self.n_shared = n_map
        if dense:
            cap = self.slam._cap
            self.acc = torch.zeros((cap, self.D), dtype=torch.float32, device=self.device)
            self.cnt = torch.zeros(cap, dtype=torch.int32, device=self.device)
        self.inst_delta = torch.zeros((4096, self.D), dtype=torch.float32, device=self.device)
        self.inst_delta_cnt = torch.zeros(4096, dtype=torch.float32, device=self.device)
        self.xchg = torch.zeros((128, 1 + self.D), dtype=torch.float32, device=self.device)     # per-step exchange rows: slot | descriptor
        self.last: Dict[str, object] = {}
        if parallel.world_size() > 1:
            # one exchange with nothing to send: every torch kernel variant of the fold is loaded here, not inside a timed
            # step (a first use costs 100-150 ms on ROCm; with several ranks loading at once, seconds were measured)
            self._exchange(torch.zeros((1, self.D), dtype=torch.float32, device=self.device), [0])
            self.inst_delta.zero_(); self.inst_delta_cnt.zero_()
            torch.cuda.synchronize()

    def _exchange(self, desc: Optional[torch.Tensor], slots: List[int]) -> None:
        """The one exchange step of the frame-sharded ranks: this keyframe's descriptor contributions as a FIXED-SIZE gather
        of the touched rows (slot | descriptor), issued by EVERY rank on EVERY step -- a rank whose frame matched nothing
        sends rows flagged -1, it never skips the collective.  Every rank folds everyone's rows into its instance table."""
        self.xchg[:, 0] = -1.0
        if desc is not None:
            k = min(desc.shape[0], self.xchg.shape[0])
            self.xchg[:k, 0] = torch.tensor(slots[:k], dtype=torch.float32).to(self.device, non_blocking=True)
            self.xchg[:k, 1:] = desc[:k]
        rows = parallel.allgather_rows(self.xchg)
        valid = rows[:, 0] >= 0
        idx = rows[:, 0].clamp(min=0).long()
        self.inst_delta.index_add_(0, idx, rows[:, 1:] * valid[:, None])
        self.inst_delta_cnt.index_add_(0, idx, valid.float())

    # ------------------------------------------------------------------ one keyframe
    def _launch_encoders(self, group: List[Frame]) -> None:
        """SAM2 image encoder and ViT forward of a group of keyframes, each as one batched forward on its side stream."""
        if self.sam is not None:
            side = torch.cuda.current_stream() if self.serial else (self.sam_stream or torch.cuda.current_stream())
            for g in group:
                if g.ready is not None:                            # the frame's upload, if it is still in flight
                    side.wait_event(g.ready)
            with torch.cuda.stream(side):
                s = self.sam.spec.image_size
                if self._sam_in is None or self._sam_in.shape[0] < len(group):
                    self._sam_in = torch.empty((len(group), 3, s, s), dtype=torch.float32, device=self.device)
                for k, g in enumerate(group):
                    self.sam.preprocess(g.rgb.permute(2, 0, 1).contiguous(), out=self._sam_in[k:k + 1])
                self.sam_out = self.sam.forward(self._sam_in[:len(group)])
        if self.prefetch:
            self.ovo.prefetch_image_features_batch([g.rgb for g in group], [g.ready for g in group if g.ready is not None],
                                                   stream=torch.cuda.current_stream() if self.serial else None)
        for g in group:
            self._encoded[g.index] = True
        self._group_first[group[0].index] = len(group)

    def step(self, f: Frame, upcoming: Optional[List[Frame]] = None) -> Dict[str, object]:
        """One keyframe.  `upcoming`: the frames that follow (only read when encoder_batch > 1: the next encoder_batch - 1 of them
        are encoded together with `f` when `f` has not been encoded yet)."""
        lib = L.load()
        self.masks.frames = {f.index: f}
        amg_pending = None
        if self.encoder_batch > 1:
            upcoming = list(upcoming or [])
            if f.index not in self._encoded:
                self._launch_encoders([f] + upcoming[:self.encoder_batch - 1])
            n_group = self._group_first.pop(f.index, 0)
            if n_group:                                            # first frame of its group: the NEXT group's encoders start now, so
                nxt = [g for g in upcoming[n_group - 1:n_group - 1 + self.encoder_batch] if g.index not in self._encoded]   # that they
                if nxt:                                            # run beside this group's tracking / pooling / fusion / queries
                    self._launch_encoders(nxt)
            self._encoded.pop(f.index, None)
            return self._step_main(f, lib, amg_pending)
        # The two encoders first: nothing of theirs depends on the map, and the map update below ends in a host sync (the
        # count of new points) behind which the host could not launch them.
        if self.sam is not None:                                   # SAM2 image encoder (masks come from the seam)
            # Independent of the tracking / descriptor work of this frame: it runs on its own HIP stream so that the two
            # kernel sequences fill each other's tails (most launches here are one or two workgroup rounds long).
            side = self.sam_stream or torch.cuda.current_stream()
            if f.ready is not None:                                # the frame's upload, if it is still in flight
                side.wait_event(f.ready)
            with torch.cuda.stream(side):
                if self.amg is not None:                           # the whole generator: encoder + 256-click decoder + filters
                    amg_pending = self.amg.generate_launch(f.rgb)
                else:
                    self.sam_out = self.sam.forward(self.sam.preprocess(f.rgb.permute(2, 0, 1).contiguous()))
        if self.prefetch:                                          # ViT tokens do not depend on the masks: start them now
            self.ovo.prefetch_image_features(f.rgb, f.ready)
        return self._step_main(f, lib, amg_pending)

    def _step_main(self, f: Frame, lib, amg_pending) -> Dict[str, object]:
        fd = [f.index, f.rgb_lr, f.depth, f.c2w]
        self.slam.track_camera(fd)
        c2w = self.slam._c2w_host[f.index]                         # host copy: no D2H for the frustum set-up
        self.slam.map(fd, c2w)
        ratio = (1.0, 1.0, self.crop_edge) if self.crop_edge else ()
        updated = self.ovo.detect_and_track_objects([f.index, f.rgb, f.depth, ratio], self.slam.get_map(), c2w)
        if updated is not None:
            self.slam.update_pcd_obj_ids(updated)
        self.ovo.compute_semantic_info()
        n = self.slam._n
        desc = getattr(self.ovo, "last_clip_embeds", None)
        if desc is not None and self.ovo.last_clip_kf == self.ovo.kf_id - 1 and desc.shape[0] > 0:
            if self.dense:
                rows = torch.tensor(self.ovo.last_mask_rows, dtype=torch.int32).to(self.device, non_blocking=True)
                L.check(lib.ovo_scatter_accum(L.ptr(self.ovo.last_point_seg), self.ovo.last_point_seg.shape[0], L.ptr(rows), rows.shape[0],
                                              L.ptr(desc), self.D, L.ptr(self.acc), L.ptr(self.cnt), L.stream()))
        else:
            desc = None
        if parallel.world_size() > 1:
            self._exchange(desc, [self.ovo.bank.slot_of[i] % 4096 for i in self.ovo.last_clip_ins_ids] if desc is not None else [])
        out: Dict[str, object] = {"n_points": n, "n_instances": len(self.ovo.objects)}
        if len(self.ovo.objects) > 0:                              # query: instances x texts, fused argmax
            table = self.ovo.get_objs_clips()
            out["sim"], out["cls"], out["conf"] = clip_utils.similarity(table, self.texts, want_argmax=True)
        if self.dense:                                             # dense query: per-point mean descriptor x texts
            _, out["dense_cls"], out["dense_conf"] = clip_utils.similarity(self.acc[:n], self.texts, cnt=self.cnt[:n], want_sim=False,
                                                                           want_argmax=True)
        if amg_pending is not None:                                # host filter + NMS + binarise: by now the statistics are long there
            self.sam_out = self.amg.generate_finish(amg_pending)
        if self.join_each_step:                                    # strict frame boundaries (tests); the stream of frames is
            self.join()                                            # otherwise software-pipelined: tail(t) || encoders(t+1)
        self.last = out
        return out

    def join(self) -> None:
        """Make the main stream wait for the SAM2 and ViT streams (everything of the frames stepped so far)."""
        for side in (self.sam_stream, self.ovo._vit_stream):
            if side is not None and side != torch.cuda.current_stream():
                torch.cuda.current_stream().wait_stream(side)

    def merge_dense(self) -> int:
        """Merge the per-GPU dense accumulators over xGMI (called once per batch of frames / before a global query).
        Only the points every rank shares -- the map all replicas started from -- are merged: each rank appends its own
        frames' points after them, so sizes beyond `n_shared` differ per rank and a collective over them would mismatch."""
        return parallel.allreduce_dense_(self.acc[:self.n_shared], self.cnt[:self.n_shared]) if self.dense else 0

    # ------------------------------------------------------------------ workload accounting
    def flops_per_frame(self, h: int, w: int) -> Dict[str, float]:
        spec = self.clip.model.spec
        crops = 1 + max(h // spec.image_size, 1) * max(w // spec.image_size, 1)
        out = {"vit": crops * spec.flops_per_image()}
        if self.sam is not None:
            out["sam2"] = self.sam.spec.flops_per_image()
        return out
