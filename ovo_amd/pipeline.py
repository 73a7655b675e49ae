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

Several GPUs (one process each): a ROUND is `world` consecutive keyframes, keyframe k of the round OWNED by rank k.
  * heavy, frame-independent work -- SAM2 encoder (+ mask generator), ViT tokens, region pooling -- runs on the owner only;
  * the order-dependent integer passes -- back-projection append (vanilla_mapper.py:81-85), cull / project / vote / instance-id
    allocation (ovo.py:255-282) -- run on EVERY rank for EVERY keyframe of the round in keyframe order against a replicated map: they
    are deterministic, so all replicas stay bit-identical with no message at all (what they need of a foreign frame is its depth, pose
    and masks: inputs every rank receives; masks produced by a rank's OWN generator (SAM2 end to end, or `mask_source`) reach the other ranks
    bit-packed through `parallel.share_masks` before the round is tracked, `_exchange_masks`);
  * the ONE exchange of a round: an all-gather of the owners' descriptors f32[<= 128, D] (KBs over xGMI / RCCL); every rank then stores and
    re-fuses them in keyframe order (`OVO._apply_semantic_plan`), so the instance tables are identical too;
  * the dense per-point accumulators are SHARDED by point (block-cyclic): each rank applies every keyframe's descriptors to its own
    rows, in keyframe order -- the merged accumulator is the concatenation of the shards and equals the one-process accumulator bit
    for bit (a floating-point all-reduce of per-GPU partial sums would not); the dense query runs on the local rows only.
tests/test_gpu_multirank.py runs two ranks on one GPU and asserts equality with the one-process run.
"""
from __future__ import annotations

import contextlib
import os
import time
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
from .entities.round_chain import RoundLauncher
from .slam.vanilla_mapper import VanillaMapper
from .utils import clip_utils, geometry_utils as G
from .utils.streams import side_stream


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
        from .utils import geometry_utils as G
        out.append(Frame(t, torch.from_numpy(rgb).to(device), torch.from_numpy(np.ascontiguousarray(rgb[e:H - e, e:W - e])).to(device),
                         G.tag_depth_range(torch.from_numpy(depth).to(device), depth), c2w, torch.from_numpy(syn.masks_to_segmap(masks)).to(device),
                         torch.from_numpy(masks).to(device)))
    return out


def _hwc(rgb: torch.Tensor) -> torch.Tensor:
    """An HWC u8 frame as the resize kernel reads it in place; anything else goes through a CHW copy."""
    return rgb if rgb.dtype == torch.uint8 and rgb.is_contiguous() else rgb.permute(2, 0, 1).contiguous()


class FramePipeline:
    SHARD_BLOCK = 4096          # points per block of the block-cyclic dense-accumulator shards
    MAX_DESC = 128              # descriptor rows a keyframe contributes to the exchange (masks per frame <= 128)

    def __init__(self, device="cuda", vit_card: str = "PE-Core-L14-336", sam_card: Optional[str] = "hiera_b+",
                 n_map: int = 1_000_000, n_text: int = 10, dense: bool = True, scale: float = 1.0, extra_capacity: int = 4_000_000,
                 seed: int = 0, depth_filter: bool = True, track_th: int = 100, sam_full: bool = False, points_per_side: int = 16,
                 encoder_batch: int = 1, k_top_views: int = 10000, emulate: Optional[tuple] = None, share_crops: bool = False,
                 own_masks: bool = False, amg_thresholds: Optional[tuple] = None, nms_score_thr: float = 0.7, min_own_masks: int = 0):
        """`emulate = (rank, world)`: ONE process does exactly what rank `rank` of a `world`-GPU job does per round -- its own keyframe's
        encoders and pooling, every keyframe's replicated passes, 1 / world of the dense rows -- with the round's all-gather replaced by a
        local stand-in (the other owners' descriptors are copies of its own rows).  A timing tool (bench.py `projection`): it measures a
        rank's round time on one GPU; its outputs are not results."""
        self.device = torch.device(device)
        self.scale = scale
        self.emulate = emulate is not None
        if self.emulate:
            self.rank, self.world = int(emulate[0]), int(emulate[1])
        else:
            self.rank = torch.distributed.get_rank() if parallel.world_size() > 1 else 0
            self.world = parallel.world_size()
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
        clip_cfg = {"embed_type": "TextRegion", "model_card": vit_card, "k_top_views": k_top_views, "fusion": "avg_pooling", "seed": seed,
                    "share_identical_crops": bool(share_crops)}
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
            kw = {} if amg_thresholds is None else {"pred_iou_thresh": float(amg_thresholds[0]), "stability_score_thresh": float(amg_thresholds[1])}
            hs = HIERA_SPECS[sam_card]
            fit = [d for d in DEC_SPECS.values() if d.hidden == hs.fpn_dim and d.image_size == hs.image_size and d.embed_size * 16 == hs.image_size]
            if not fit:
                raise ValueError(f"no SAM2 mask decoder spec for encoder {sam_card} (FPN width {hs.fpn_dim}, input {hs.image_size})")
            self.amg = HipSam2AutomaticMaskGenerator(self.sam, HipSamDecoder(fit[0], None, self.device, seed), points_per_side=points_per_side, **kw)
        # `own_masks` (with sam_full): the masks SAM2's generator produced for a keyframe drive ITS tracking -- the reference's default path
        # (mask_generator.py:102-120: generate -> masks_update -> mask2segmap -> ovo.py:182-324), also on one GPU; off: they are produced and
        # measured but tracking consumes the masks the frame carries (the precomputed-mask seam, mask_generator.py:94-95)
        self.own_masks = bool(own_masks) and self.amg is not None
        self.nms_score_thr = float(nms_score_thr)
        # fewer surviving masks than this: the keyframe is tracked with the masks the frame carries (AFTER the generator, its mask NMS and its seg
        # map ran: the dependency and the cost are the real chain's).  Random-init SAM2 weights produce empty or whole-image masks of which the mask
        # NMS keeps ~3 (tools/bin/amg_probe.py), so a throughput run sets this to keep a realistic mask count in the tracker (bench.py --sam-own-masks)
        self.min_own_masks = int(min_own_masks)
        self.own_fallbacks = 0
        self.own_mask_counts: List[int] = []
        # encoder look-ahead: the two encoders of `encoder_batch` consecutive keyframes (of this rank) run as ONE batched forward each
        # (step() is handed the upcoming frames).  Nothing of the encoders depends on the map, and the reference itself computes a
        # keyframe's descriptors kf_queue_delay = 10 keyframes late (ovo.yaml:53), so this changes no result -- only the GEMM height.
        self.encoder_batch = max(1, int(encoder_batch))
        self._encoded: Dict[int, bool] = {}
        self.serial = False                                        # measurement only: both encoders on the caller's stream
        self._group_first: Dict[int, int] = {}
        self._sam_in = None
        self._sam_by_frame: Dict[int, tuple] = {}
        self.sam_frame = None                                      # SAM2 features (f0, f1, f2) of the keyframe this rank last stepped
        self.prefetch = not os.environ.get("OVO_NO_PREFETCH")
        self.join_each_step = bool(os.environ.get("OVO_JOIN_EACH_STEP"))
        self.sam_stream = side_stream(self.device, "OVO_SAM_CUS", int(os.environ.get("OVO_SAM_PRIORITY", "0"))) if (sam_card and not os.environ.get("OVO_SAM_SAME_STREAM")) else None
        self.D = self.clip.clip_dim
        self.texts = torch.from_numpy(syn.unit_vectors(n_text, self.D, seed=seed + 7)).to(self.device)
        self.dense = dense
        self.exchange_ms = 0.0                                     # host wall time spent in the rounds' collectives (bench.py reports it)
        self.exchanges = 0
        self.exchange_events: list = []                            # (start, end) hipEvents around every round's all-gather
        if dense:
            cap = self.slam._cap
            blocks = -(-cap // self.SHARD_BLOCK)
            self.rows_local = -(-blocks // self.world) * self.SHARD_BLOCK if self.world > 1 else cap    # rows of THIS rank's shard
            self.acc = torch.zeros((self.rows_local, self.D), dtype=torch.float32, device=self.device)
            self.cnt = torch.zeros(self.rows_local, dtype=torch.int32, device=self.device)
            # The dense class / confidence map stays RESIDENT: a keyframe changes the accumulators of the points it matched (10-20 % of
            # the map) and only those rows can change class, so the scatter pass emits their indices and the query re-evaluates just
            # them (`ovo_similarity_rows`) -- bit-identical to re-querying all rows (tests/test_gpu_pipeline.py), a fraction of the 5 GB
            # stream.  Initial state = the query of the empty accumulators, computed once here over the whole capacity.
            self.incremental_query = self.D % 16 == 0 and not os.environ.get("OVO_DENSE_FULL_QUERY")
            if self.incremental_query:
                _, self.dense_cls, self.dense_conf = clip_utils.similarity(self.acc, self.texts, cnt=self.cnt, want_sim=False, want_argmax=True)
                self.touched = torch.empty(self.rows_local, dtype=torch.int32, device=self.device)
                self.n_touched = torch.zeros(2, dtype=torch.int32, device=self.device)     # two counters, used alternately
                self._touch_parity = 0
                # round 6: the tracking chain lists the points its masks cover and ONE launch accumulates and re-queries them (`ovo_scatter_accum_query`);
                # the scan + apply + query launches above remain for keyframes tracked on the host path and for OVO_NO_FUSED_SCATTER=1
                if not os.environ.get("OVO_NO_FUSED_SCATTER") and n_text <= 16 and n_text * self.D * 4 <= 96 * 1024:
                    self.ovo.hit_shard = (self.rank, self.world, self.SHARD_BLOCK)
        self.last: Dict[str, object] = {}
        # Masks from this rank's OWN generator (SAM2 end to end, or an injected `mask_source(frame) -> (seg_map, masks)`): the owner of a
        # keyframe produces them, `parallel.share_masks` carries them bit-packed to the replicas, which track with exactly those bits.
        self.mask_source = None
        self.mask_exchanges = 0
        self._chains: Dict[int, list] = {}                         # first frame index of a round -> its queued chains (software pipelining)
        self._round_seq: Dict[int, int] = {}                       # first frame index of a round -> sequence number of its last map step
        # result rings for two rounds in flight (this one being read, the next one pre-queued) plus margin: a full ring would make the
        # launchers wait for result blocks of steps that are built but not launched yet
        self.slam.ring_slots(max(64, 4 * self.world + 8))
        self.ovo._track_ring_slots = max(32, 4 * self.world + 8)
        self.pipeline_rounds = not os.environ.get("OVO_NO_ROUND_PIPELINE")
        self.round_launcher = RoundLauncher(self.device)           # one persistent launch per round instead of ~12 launches per keyframe
        # The chains of consecutive keyframes depend on each other (device-resident map size / instance ids) and so do the keyframes' tails
        # (descriptor store, fusion, dense scatter + query); a chain and a tail of DIFFERENT keyframes do not.  On one stream they queue
        # behind each other -- ~25 small dependent launches per keyframe, each waiting for a free CU among the encoders' workgroups -- so
        # the chains get a stream of their own and overlap the tails (events: chain k -> tail k; the tail's stream -> the chain's inputs).
        self.chain_stream = None if os.environ.get("OVO_NO_CHAIN_STREAM") else torch.cuda.Stream(device=self.device, priority=int(os.environ.get("OVO_CHAIN_PRIORITY", "0")))
        if self.world > 1:
            self.xchg = torch.zeros((self.MAX_DESC, self.D), dtype=torch.float32, device=self.device)
            # (rows beyond a keyframe's descriptors keep stale values: every rank knows the counts from the replicated plans and never reads them)
            self._gather(self.xchg)                                # first use of the collective (and of its kernels) outside any timed step
            torch.cuda.synchronize()

    def exchange_device_ms(self, last: Optional[int] = None) -> Optional[float]:
        """Mean device time of the rounds' all-gathers (the newest `last` of them), from the hipEvents around each; None before the first.
        Call after a synchronize."""
        ev = self.exchange_events[-last:] if last else self.exchange_events
        return sum(a.elapsed_time(b) for a, b in ev) / len(ev) if ev else None

    def drain(self) -> None:
        """End of stream: a round that was pre-queued (software pipelining, `step_round(..., upcoming=...)`) but never stepped has already
        advanced the map and the per-point instance ids on the device.  Read its result blocks and do the host bookkeeping -- the instances
        and keyframe queue then match the map again; its keyframes get no descriptors (call `OVO.complete_semantic_info()` for those)."""
        for first in sorted(self._chains):
            for p in self._chains.pop(first):
                self.ovo.detect_and_track_finish(p)
            self._round_seq.pop(first, None)
        self.slam.settle()

    def _gather(self, t: torch.Tensor) -> torch.Tensor:
        if self.emulate:                                           # stand-in with the collective's output shape and one device copy
            return t[None].expand(self.world, *t.shape).contiguous()
        return parallel.allgather(t)

    # ------------------------------------------------------------------ encoders
    def prime(self, H: int, W: int) -> None:
        """Set-up, not work of any frame: one batched forward of both encoders at the full look-ahead width on blank H x W frames, once per
        token slot, so that workspaces / slots have their final size (device allocations, first launches of the batch-sized kernel
        variants) before the first real group -- which may arrive inside a timed region after a shorter warm-up group."""
        if self.encoder_batch <= 1:
            return
        for _ in range(2):                                         # two token slots (double-buffered batches)
            blank = [Frame(-1 - k, torch.zeros((H, W, 3), dtype=torch.uint8, device=self.device), None, None, None, None, None)
                     for k in range(self.encoder_batch)]
            self._launch_encoders(blank)
            torch.cuda.synchronize()
            for f in blank:
                self.ovo._prefetched_batch.pop(id(f.rgb), None)
            for slot in (self.ovo._batch_slots or []):
                slot["left"], slot["free"] = 0, None
        self._sam_by_frame.clear(); self._encoded.clear(); self._group_first.clear()

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
                self.sam.preprocess_batch([_hwc(g.rgb) for g in group], out=self._sam_in[:len(group)])      # the HWC frames are read in place, one launch
                self.sam_out = self.sam.forward(self._sam_in[:len(group)])
                for k, g in enumerate(group):                      # a frame's features = slice k of the batched output
                    self._sam_by_frame[g.index] = (self.sam_out, k)
        if self.prefetch:
            self.ovo.prefetch_image_features_batch([g.rgb for g in group], [g.ready for g in group if g.ready is not None],
                                                   stream=torch.cuda.current_stream() if self.serial else None)
        for g in group:
            self._encoded[g.index] = True
        self._group_first[group[0].index] = len(group)

    def _encoders_for(self, f: Frame, mine_upcoming: List[Frame]):
        """Start the encoders `f` needs (and, with look-ahead batching, those of this rank's next frames).  Returns the pending mask
        generator call of `f`, if the pipeline runs SAM2 end to end."""
        if self.encoder_batch > 1:
            def width(rem):                                        # full groups, then what is left (measured against balanced 7 + 7 + 6 and a
                return min(self.encoder_batch, rem)                # tapered tail 8 + 8 + 6 + 2 on one box: 305 / 304 / 295 frames/s at 20 steps)
            if f.index not in self._encoded:
                self._launch_encoders([f] + mine_upcoming[:width(1 + len(mine_upcoming)) - 1])
            n_group = self._group_first.pop(f.index, 0)
            if n_group:                                            # first frame of its group: the NEXT group's encoders start now, so
                rest = [g for g in mine_upcoming[n_group - 1:] if g.index not in self._encoded]                                  # that they
                nxt = rest[:width(len(rest))]                      # run beside this group's tracking / pooling / fusion / queries
                if nxt:
                    self._launch_encoders(nxt)
            self._encoded.pop(f.index, None)
            if self.amg is None:
                return None
            # SAM2 end to end: the encoder ran batched for the group; decoder + generator filters run per frame on the SAM2 stream
            hit = self._sam_by_frame.get(f.index)
            outs, k = hit
            emb = {"image_embed": outs[2][k:k + 1], "high_res_feats": (outs[0][k:k + 1], outs[1][k:k + 1])}
            side = torch.cuda.current_stream() if self.serial else (self.sam_stream or torch.cuda.current_stream())
            with torch.cuda.stream(side):
                return self.amg.generate_launch(f.rgb, embeddings=emb)
        amg_pending = None
        # The two encoders first: nothing of theirs depends on the map, and the map update ends in a host sync (the count of new
        # points) behind which the host could not launch them.
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
                    self.sam_out = self.sam.forward(self.sam.preprocess(_hwc(f.rgb)))
                    self._sam_by_frame[f.index] = (self.sam_out, 0)
        if self.prefetch:                                          # ViT tokens do not depend on the masks: start them now
            self.ovo.prefetch_image_features(f.rgb, f.ready)
        return amg_pending

    # ------------------------------------------------------------------ one keyframe (one process) / one round (several)
    def step(self, f: Frame, upcoming: Optional[List[Frame]] = None) -> Dict[str, object]:
        """One keyframe on one GPU.  `upcoming`: the frames that follow (only read when encoder_batch > 1: the next ones are encoded
        together with `f` when `f` has not been encoded yet)."""
        if self.world > 1:
            raise L.OvoHipError("several ranks step whole rounds: use step_round()")
        return self.step_round([f], upcoming)

    def step_round(self, group: List[Frame], upcoming: Optional[List[Frame]] = None) -> Dict[str, object]:
        """`world` consecutive keyframes, keyframe k owned by rank k (one process: a round is one keyframe).  `upcoming`: the frames
        after the round, in order (their owners follow the same k = position % world rule)."""
        lib = L.load()
        if len(group) != self.world:
            raise L.OvoHipError(f"a round is {self.world} keyframes, got {len(group)}")
        upcoming = list(upcoming or [])
        mine = group[self.rank]
        self.masks.frames = {f.index: f for f in group}
        amg_pending = self._encoders_for(mine, upcoming[self.rank::self.world])
        hit = self._sam_by_frame.pop(mine.index, None)
        if hit is not None:
            self.sam_frame = tuple(t[hit[1]:hit[1] + 1] for t in hit[0])
        if self.mask_source is not None or (self.amg is not None and (self.world > 1 or self.own_masks)):
            self._exchange_masks(group, amg_pending)
            amg_pending = None
        # ---- the order-dependent passes, for every keyframe of the round, on every rank (replicated map and tracker).  The whole
        # round is QUEUED first -- map update and tracking chain of every keyframe, sizes and instance ids device-resident
        # (`ovo_map_step` / `ovo_track_step`) -- then finished in order: the host bookkeeping of keyframe k runs while the device
        # works on k + 1 ..., and nothing on the device ever waits for the host.
        plans, segs = [], []
        ratio = (1.0, 1.0, self.crop_edge) if self.crop_edge else ()
        native = not self.ovo.config.get("log", False) and all(self.ovo._native_ok(self.masks.frames[f.index].masks) for f in group)
        if native:
            pend = self._chains.pop(group[0].index, None) or self._launch_chains(group, ratio)
            # software pipelining of rounds: the NEXT round's chains are queued now, before this round's results are read -- while the
            # host does this round's bookkeeping, pooling, exchange and fusion the device already works on the next round's tracking,
            # and vice versa (queue -> wait -> bookkeeping in one round leaves host and device waiting for each other in turn)
            nxt = upcoming[:self.world]
            if self.pipeline_rounds and len(nxt) == self.world and self.mask_source is None and self.amg is None \
                    and all(self.ovo._native_ok(f.masks) for f in nxt):
                self.masks.frames.update({f.index: f for f in nxt})
                self._chains[nxt[0].index] = self._launch_chains(nxt, ratio)
            for k, p in enumerate(pend):                           # (assignment happened in place in the mapper's buffer; only the owner of a
                self.ovo.detect_and_track_finish(p, want_maps=(k == self.rank))     # keyframe reads its kept binary maps: `_extract_clip` below)
                plans.append(self.ovo._plan_semantic_info() if len(self.ovo.keyframes_queue) > 0 else None)
                segs.append((self.ovo.last_point_seg, self.ovo.last_mask_rows, self.ovo.last_hits))
            # the map's size after this round: from the round's last map step (a keyframe without masks reports none through the tracker,
            # and a pre-queued NEXT round may already have moved the mapper's own count past it)
            seq = self._round_seq.pop(group[0].index, 0)
            n = self.slam.size_after(seq) if seq else self.ovo.last_n_points
        else:
            for f in group:
                fd = [f.index, f.rgb_lr, f.depth, f.c2w]
                self.slam.track_camera(fd)
                c2w = self.slam._c2w_host[f.index]
                self.slam.map(fd, c2w)
                updated = self.ovo.detect_and_track_objects([f.index, f.rgb, f.depth, ratio], self.slam.get_map(), c2w)
                if updated is not None:
                    self.slam.update_pcd_obj_ids(updated)
                plans.append(self.ovo._plan_semantic_info() if len(self.ovo.keyframes_queue) > 0 else None)
                segs.append((self.ovo.last_point_seg, self.ovo.last_mask_rows, self.ovo.last_hits))
            n = self.slam._n
        # ---- descriptors of the keyframe this rank owns, then the round's one exchange
        plan = plans[self.rank]
        desc_mine = self.ovo._extract_clip(plan["image"], plan["binary_maps"]) if plan is not None else None
        if self.world > 1:
            for p in plans:                                        # the plans are replicated: every rank sees an overflow, none is left in the collective
                if p is not None and len(p["matched_ins_ids"]) > self.MAX_DESC:
                    raise L.OvoHipError(f"{len(p['matched_ins_ids'])} descriptors in one keyframe: raise FramePipeline.MAX_DESC")
            t0 = time.perf_counter()
            if desc_mine is not None:
                self.xchg[:desc_mine.shape[0]].copy_(desc_mine)
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) if len(self.exchange_events) < 4096 else None
            if ev:
                ev[0].record()
            gathered = self._gather(self.xchg)                     # [world, MAX_DESC, D]: every owner's descriptors, rank-major = keyframe order
            if ev:                                                 # device time of the collective itself (hipEvents on the stream it runs on)
                ev[1].record()
                self.exchange_events.append(ev)
            descs = [gathered[k, :len(p["matched_ins_ids"])] if p is not None else None for k, p in enumerate(plans)]
            self.exchange_ms += 1e3 * (time.perf_counter() - t0)
            self.exchanges += 1
        else:
            descs = [desc_mine]
        # ---- every rank: store + re-fuse in keyframe order (identical instance tables), dense accumulate on its own rows
        # (the instance table is read after the round only: the round's running-sum fusions are one launch, the mask-row lists one upload)
        self.ovo._apply_semantic_plans([(p, d) for p, d in zip(plans, descs) if p is not None])
        live = [k for k, (p, d) in enumerate(zip(plans, descs)) if p is not None and self.dense and d.shape[0] > 0]
        all_rows = torch.tensor([r for k in live for r in segs[k][1]], dtype=torch.int32).to(self.device, non_blocking=True) if live else None
        row_off = 0
        for k, (p, d, (point_seg, mask_rows, hits)) in enumerate(zip(plans, descs, segs)):
            if p is None:
                continue
            if self.dense and d.shape[0] > 0:
                rows = all_rows[row_off:row_off + len(mask_rows)]
                row_off += len(mask_rows)
                if hits is not None and self.incremental_query:    # one launch: accumulate the listed rows and re-evaluate exactly them
                    n_list = hits.numel() - 4
                    L.check(lib.ovo_scatter_accum_query(L.ptr(hits), hits[n_list:].data_ptr(), min(point_seg.shape[0], self.rows_local), L.ptr(point_seg),
                                                        L.ptr(rows), rows.shape[0], L.ptr(d), self.D, L.ptr(self.acc), L.ptr(self.cnt), self.rank, self.world,
                                                        self.SHARD_BLOCK, L.ptr(self.texts), self.texts.shape[0], 0, 0.0, 0.0, 0.0,
                                                        L.ptr(self.dense_cls), L.ptr(self.dense_conf), L.stream()))
                    continue
                touched, n_cur, n_nxt = None, None, None
                if self.incremental_query:
                    k = self._touch_parity
                    self._touch_parity ^= 1
                    touched, n_cur, n_nxt = L.ptr(self.touched), self.n_touched[k:].data_ptr(), self.n_touched[k ^ 1:].data_ptr()
                L.check(lib.ovo_scatter_accum_touched(L.ptr(point_seg), point_seg.shape[0], L.ptr(rows), rows.shape[0], L.ptr(d), self.D,
                                                      L.ptr(self.acc), L.ptr(self.cnt), touched, n_cur, n_nxt, self.rank, self.world,
                                                      self.SHARD_BLOCK, L.stream()))
                if self.incremental_query:                         # only the rows this keyframe changed can change class
                    L.check(lib.ovo_similarity_rows(L.ptr(self.acc), 0, touched, n_cur, min(point_seg.shape[0], self.rows_local), self.D,
                                                    L.ptr(self.texts), self.texts.shape[0], L.ptr(self.cnt), 0, 0.0, 0.0, 0.0,
                                                    L.ptr(self.dense_cls), L.ptr(self.dense_conf), L.stream()))
        out: Dict[str, object] = {"n_points": n, "n_instances": len(self.ovo.objects)}
        if len(self.ovo.objects) > 0:                              # query: instances x texts, fused argmax
            table = self.ovo.get_objs_clips()
            out["sim"], out["cls"], out["conf"] = clip_utils.similarity(table, self.texts, want_argmax=True)
        if self.dense:
            nl = self.local_rows(n)
            if self.incremental_query:                             # the resident map, patched above for the rows the round changed
                out["dense_cls"], out["dense_conf"] = self.dense_cls[:nl], self.dense_conf[:nl]
            else:                                                  # dense query: per-point mean descriptor x texts, every (local) row
                _, out["dense_cls"], out["dense_conf"] = clip_utils.similarity(self.acc[:nl], self.texts, cnt=self.cnt[:nl], want_sim=False,
                                                                               want_argmax=True)
        if amg_pending is not None:                                # host filter + NMS + binarise: by now the statistics are long there
            self.sam_out = self.amg.generate_finish(amg_pending)
        if self.join_each_step:                                    # strict frame boundaries (tests); the stream of frames is
            self.join()                                            # otherwise software-pipelined: tail(t) || encoders(t+1)
        self.last = out
        return out

    # ------------------------------------------------------------------ owner -> replica masks
    def _launch_chains(self, group: List[Frame], ratio) -> list:
        """Queue map update + tracking chain of every keyframe of a round -- no host round trip -- as ONE launch (`ovo_round_chain`:
        persistent workgroups walking through all passes of all keyframes); a round that call does not cover goes keyframe by
        keyframe (`ovo_map_step` / `ovo_track_step`)."""
        G.prepare_frame_cameras([(f.depth, f.c2w) for f in group], self.slam._K_host)
        side = self.chain_stream
        # capacity for the WHOLE round before its first deferred step exists: growing re-allocates the map's buffers, and the steps built
        # below (and the previous round's, still on the chain stream) hold their addresses
        self.slam.reserve_round([tuple(f.depth.shape) for f in group], sync=(side,))
        maps, tracks, pend = [], [], []
        # Everything the chains touch is produced ON their stream (the masks' working copy, the per-keyframe buffers): nothing of the chain
        # waits for the main stream, whose queue holds the previous rounds' tails (with a main -> chain dependency the two streams took turns:
        # tail(r - 1) -> chains(r + 1) -> tail(r + 1) ..., 4.5 ms per round).  The other direction is an event: chain k -> tail k.
        with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
            round_seq = 0                                          # sequence number of the round's LAST map step (0: none was built)
            for f in group:
                fd = [f.index, f.rgb_lr, f.depth, f.c2w]
                self.slam.track_camera(fd)
                c2w = self.slam._c2w_host[f.index]                 # host copy: no D2H for the frustum set-up
                if f.ready is not None and side is not None:
                    side.wait_event(f.ready)                       # the frame's upload, if it is still in flight
                m = self.slam.map_launch(fd, c2w, defer=True)
                if m is not None:
                    round_seq = self.slam.last_seq
                p = self.ovo.detect_and_track_launch([f.index, f.rgb, f.depth, ratio], self.slam, c2w, defer=True)
                pend.append(p)
                if m is not None or p is not None:
                    maps.append(m if m is not None else L.MapStep())
                    tracks.append(p["step"] if p is not None else L.TrackStep())
            # the map's size after the round = result block of its last map step; a round in which no keyframe had valid depth built none
            # (the mapper's last_seq then names an OLDER round's slot, possibly reused): the consumer falls back to the tracker's count
            self._round_seq[group[0].index] = round_seq
            if maps:
                self.round_launcher.launch(maps, tracks, None)     # (the current stream IS the chain stream here)
            self.slam.launched()
            if maps:
                if side is not None:
                    done = torch.cuda.Event()
                    done.record(side)
                    for p in pend:
                        if p is not None:
                            p["done"] = done
        return pend

    def _own_masks(self, mine: Frame, amg_pending):
        """(seg_map, masks) of the keyframe this rank owns from ITS generator; (None, None) when it kept no mask."""
        if self.mask_source is not None:
            seg, masks = self.mask_source(mine)
            return (seg, masks) if masks is not None and masks.shape[0] > 0 else (None, None)
        from .utils import segment_utils
        r = self.amg.generate_finish(amg_pending)                  # filter + box NMS (mask_generator.py:113)
        self.sam_out = r
        masks = r["masks"]
        if masks.shape[0] == 0:
            self.own_mask_counts.append(0)
            return None, None
        keep = segment_utils.masks_update_device(masks, r["predicted_iou"], r["stability_score"], iou_thr=0.8, score_thr=self.nms_score_thr, inner_thr=0.5)
        self.own_mask_counts.append(int(keep.numel()))
        seg, ordered = segment_utils.mask2segmap_device(masks.index_select(0, keep.to(masks.device)), r["stability_score"][keep.numpy()])
        if keep.numel() < self.min_own_masks:
            self.own_fallbacks += 1
            return None, None
        return seg, ordered

    def _exchange_masks(self, group: List[Frame], amg_pending) -> None:
        """Every keyframe of the round is tracked on every rank with the masks its OWNER's generator produced (mask_generator.py:102-120 runs
        on the owner only).  A keyframe whose owner kept no mask falls back to the masks the frame carries (the precomputed-mask seam,
        mask_generator.py:94-95) -- with random-init SAM2 weights that is every keyframe, stated in bench.py's help."""
        from .utils import segment_utils
        mine = group[self.rank]
        H, W = mine.rgb.shape[:2]
        _, masks = self._own_masks(mine, amg_pending)
        if self.world > 1:
            shared = parallel.share_masks(masks, H * W, self.device, gather=self._gather)
            self.mask_exchanges += 1
        else:
            shared = [torch.empty((0, H * W), dtype=torch.uint8, device=self.device) if masks is None else
                      (masks.view(torch.uint8) if masks.dtype == torch.bool else masks).reshape(masks.shape[0], -1)]
        for f, u in zip(group, shared):
            if u.shape[0] == 0:
                continue                                           # the frame's own (seam) masks stay
            seg_map = torch.empty((H, W), dtype=torch.int32, device=self.device)      # mask2segmap's painting of the received bits
            L.check(L.load().ovo_paint_segmap(L.ptr(u), u.shape[0], H * W, L.ptr(seg_map), L.stream()))
            self.masks.frames[f.index] = Frame(f.index, f.rgb, f.rgb_lr, f.depth, f.c2w, seg_map, u.view(torch.bool).reshape(-1, H, W), f.ready)
        # The received masks and their seg maps were produced on THIS stream (collective copy-back, unpack, paint); the round's chains run on the
        # chain stream, which by design waits for nothing of the main stream (`_launch_chains`: its other inputs are resident frame tensors).  These
        # are not: without this edge the chain's working copy could read a block the unpack had not written yet -- found by the world-8 one-GPU
        # run, where eight processes' kernels interleave (wrong descriptors for the last round's instances in 2 of 3 runs; never seen at world 2).
        if self.chain_stream is not None:
            self.chain_stream.wait_stream(torch.cuda.current_stream())

    # ------------------------------------------------------------------ dense shards
    def local_rows(self, n: int) -> int:
        """Rows of this rank's shard that hold points of a map with n points (block-cyclic, blocks of SHARD_BLOCK points)."""
        if self.world == 1:
            return n
        B, R, r = self.SHARD_BLOCK, self.world, self.rank
        full, rem = divmod(n, B)                                   # `full` complete blocks, then one of `rem` points
        mine = (full - r + R - 1) // R if full > r else 0          # complete blocks owned by r
        return mine * B + (rem if full % R == r else 0)

    def gather_dense(self, n: Optional[int] = None):
        """The whole dense state on every rank, in point order: (acc f32[n, D], cnt i32[n], cls i64[n], conf f32[n]).  The concatenation
        of the shards -- no arithmetic, so it equals the one-process accumulators bit for bit.  A map-sized collective: for export /
        tests, never inside the keyframe loop (queries run on the shards)."""
        n = self.slam._n if n is None else n
        if self.world == 1:
            inc = getattr(self, "incremental_query", False)
            return self.acc[:n], self.cnt[:n], self.dense_cls[:n] if inc else None, self.dense_conf[:n] if inc else None
        B, R = self.SHARD_BLOCK, self.world
        nb = -(-n // B)
        per = -(-nb // R)                                          # blocks per rank (the last rank's may be short / absent)
        rows = per * B

        def merge(local: torch.Tensor) -> torch.Tensor:
            g = self._gather(local[:rows].contiguous())            # [R, per * B, ...]
            g = g.reshape(R, per, B, *local.shape[1:]).transpose(0, 1).reshape(per * R * B, *local.shape[1:])   # block b = (b // R, b % R)
            return g[:n]
        return merge(self.acc), merge(self.cnt), merge(self.dense_cls) if self.incremental_query else None, \
            merge(self.dense_conf) if self.incremental_query else None

    def join(self) -> None:
        """Make the main stream wait for the SAM2 and ViT streams (everything of the frames stepped so far)."""
        for side in (self.sam_stream, self.ovo._vit_stream, self.chain_stream):
            if side is not None and side != torch.cuda.current_stream():
                torch.cuda.current_stream().wait_stream(side)

    # ------------------------------------------------------------------ workload accounting
    def flops_per_frame(self, h: int, w: int) -> Dict[str, float]:
        spec = self.clip.model.spec
        crops = 1 + max(h // spec.image_size, 1) * max(w // spec.image_size, 1)
        out = {"vit": crops * spec.flops_per_image()}
        if self.sam is not None:
            out["sam2"] = self.sam.spec.flops_per_image()
        return out
