"""GT-pose point-map builder on MI355X -- the "back-projection" half of the hot path.

Host-side mirror of the reference's `ovo/slam/vanilla_mapper.py:VanillaMapper` (the canonical SLAM
duck-type `OVOSemMap.run` drives: track_camera / get_c2w / map / get_map / update_pcd_obj_ids / ...).
The map lives in capacity-doubling device buffers (the reference re-allocates the whole map with
`torch.vstack` on every mapped frame, vanilla_mapper.py:81-85); per frame ONE C call (`ovo_map_step`) queues
  1. the explained-pixel pass -- cull + project + depth-test every map point, mark explained pixels  (:56-61)
  2. the back-projection      -- erode, [::2,::2] subsample, unproject, c2w, ordered append           (:62-85)
and the map's size / next point id advance in DEVICE memory: `map_launch` returns without a host round trip, the count of
appended points arrives in a pinned result block that `settle` (or any read of `_n` / `max_id` / `pcd`) waits for.  `map`, the
reference's method, is launch + settle.
"""
from __future__ import annotations

from collections import deque
from typing import Any, Dict, List, Tuple

import numpy as np
import torch

from .. import _lib as L
from ..utils import geometry_utils as G


class VanillaMapper:
    """Same constructor and methods as the reference class (vanilla_mapper.py:8-136)."""

    def __init__(self, config: dict, cam_intrinsics: torch.Tensor) -> None:
        self.cam_intrinsics = cam_intrinsics
        self.config = config
        self.device = config.get("device", "cuda")
        mapping = config.get("mapping", {})
        self.max_frame_points = mapping.get("max_frame_points", 1e5)
        self.match_distance_th = 0.03                      # vanilla_mapper.py:17
        self._max_id = 0
        self.estimated_c2ws: Dict[int, torch.Tensor] = {}
        self._c2w_host: Dict[int, torch.Tensor] = {}
        self.kfs: Dict[int, Dict[str, Any]] = {}
        self.map_updated = False
        self.k_pooling = int(mapping.get("k_pooling", 3))
        if self.k_pooling not in (1, 3):
            raise NotImplementedError("k_pooling must be 1 or 3 (the reference default is 3)")
        self.downscale = int(mapping.get("downscale_res", 2))   # sic: the yaml key `downscale_ratio` is never read (:32)
        self._K_host = G._cpu32(cam_intrinsics).contiguous()
        self._K9 = (L.C.c_float * 9)(*self._K_host.reshape(-1).tolist())
        self._n_known = 0                                  # exact when nothing is in flight
        self._pending: deque = deque()                     # (sequence number, sub-sampled pixels) of map_launch calls not read back yet
        self._state = torch.zeros(4, dtype=torch.int64, device=self.device)       # {n, next point id, flags, ticket}
        self._ring = L.PinnedRing(4, np.int64, 64)
        self._deferred = 0                                 # map steps built with defer=True and not handed to a launcher yet
        self.last_seq = 0                                  # sequence number of the newest map step (its result block holds the map's size after it)
        self._explained = None
        self._ws = None
        self._keep: deque = deque(maxlen=64)
        self._cap = 0
        self._xyz = self._ids = self._ins = self._rgb = None
        self._reserve(1 << 16)

    # ------------------------------------------------------------------ size of the map (device-resident, mirrored lazily)
    def settle(self) -> None:
        """Read back the calls in flight: afterwards `_n` / `max_id` are exact.  Waits only for what was queued."""
        while self._pending:
            r = self._ring.wait(self._pending.popleft()[0])
            self._n_known, self._max_id = int(r[2]), int(r[3])

    def reap(self) -> None:
        """`settle` for the calls that have ALREADY finished (no wait): keeps the ring short in a pipelined stream of keyframes."""
        while self._pending and self._ring.done(self._pending[0][0]):
            r = self._ring.wait(self._pending.popleft()[0])
            self._n_known, self._max_id = int(r[2]), int(r[3])

    @property
    def _n_upper(self) -> int:
        """>= the size the map has when the queued calls have run (a frame appends at most its sub-sampled pixels)."""
        return self._n_known + sum(p[1] for p in self._pending)

    @property
    def _n(self) -> int:
        if self._pending:
            self.settle()
        return self._n_known

    @_n.setter
    def _n(self, value: int) -> None:
        self.settle()
        self._n_known = int(value)

    @property
    def max_id(self) -> int:
        if self._pending:
            self.settle()
        return self._max_id

    @max_id.setter
    def max_id(self, value: int) -> None:
        self.settle()
        self._max_id = int(value)

    def map_ref(self, n_exact=None) -> "L.MapRef":
        """ovo_map_ref_t of the live buffers: the host's exact size when nothing is in flight, else "read the device state"."""
        known = not self._pending
        return L.MapRef(self._xyz.data_ptr(), self._ids.data_ptr(), self._ins.data_ptr(), self._rgb.data_ptr(), self._cap,
                        self._state.data_ptr(), self._n_known if known else -1, self._max_id if known else -1)

    # ------------------------------------------------------------------ storage
    def _reserve(self, cap: int) -> None:
        if cap <= self._cap:
            return
        new_cap = max(cap, 2 * self._cap)
        dev = self.device
        xyz = torch.empty((new_cap, 3), dtype=torch.float32, device=dev)
        ids = torch.empty((new_cap,), dtype=torch.int32, device=dev)
        ins = torch.empty((new_cap,), dtype=torch.int32, device=dev)
        rgb = torch.empty((new_cap, 3), dtype=torch.uint8, device=dev)
        if self._n:
            xyz[:self._n].copy_(self._xyz[:self._n])
            ids[:self._n].copy_(self._ids[:self._n])
            ins[:self._n].copy_(self._ins[:self._n])
            rgb[:self._n].copy_(self._rgb[:self._n])
        self._xyz, self._ids, self._ins, self._rgb, self._cap = xyz, ids, ins, rgb, new_cap

    def reserve(self, cap: int) -> None:
        self._reserve(cap)

    def sub_pixels(self, h: int, w: int) -> int:
        """Upper bound of the points one frame of h x w depth pixels appends (vanilla_mapper.py:67-68: every `downscale`-th pixel)."""
        ds = self.downscale
        return ((h + ds - 1) // ds) * ((w + ds - 1) // ds)

    def reserve_round(self, shapes, sync=None) -> None:
        """Capacity for a whole round of frames (`shapes` = their depth maps' (h, w)) BEFORE its first deferred step is built: growing the
        map re-allocates the buffers, and steps built earlier in the round (map and tracking) hold the old addresses.  `sync`: streams whose
        queued work still reads / writes the old buffers (the chain stream); they are drained first -- growth is rare (capacity doubles)."""
        extra = sum(self.sub_pixels(h, w) for h, w in shapes)
        if self._n_upper + extra <= self._cap:
            return
        if self._deferred:
            raise L.OvoHipError("reserve_round: deferred steps outstanding")
        for st in (sync or ()):
            if st is not None:
                st.synchronize()
        self.settle()
        torch.cuda.current_stream().synchronize()
        self._reserve(self._n_known + extra)
        torch.cuda.current_stream().synchronize()          # the copies into the new buffers, before any other stream touches them

    def launched(self, n: int = None) -> None:
        """The caller handed its deferred steps to a launcher."""
        self._deferred = 0 if n is None else max(0, self._deferred - n)

    def ring_slots(self, slots: int) -> None:
        """At least `slots` result blocks (rounds of N keyframes, two rounds in flight: 2 N + margin); only while nothing is in flight."""
        if slots > self._ring.slots:
            self.settle()
            self._ring = L.PinnedRing(4, np.int64, slots)

    def size_after(self, seq: int) -> int:
        """The map's size after the map step `seq` (its result block; the step must have run)."""
        return int(self._ring.wait(seq)[2])

    @property
    def pcd(self) -> torch.Tensor:
        return self._xyz[:self._n]

    @property
    def pcd_ids(self) -> torch.Tensor:
        return self._ids[:self._n].unsqueeze(1)

    @property
    def pcd_obj_ids(self) -> torch.Tensor:
        return self._ins[:self._n].unsqueeze(1)

    @property
    def pcd_colors(self) -> torch.Tensor:
        return self._rgb[:self._n]

    # ------------------------------------------------------------------ frame-callback API
    def track_camera(self, frame_data: List[Any]) -> None:
        frame_id, c2w = frame_data[0], frame_data[3]
        if not np.isfinite(c2w).all():                     # skip NaN / Inf poses (:41-42)
            return
        host = torch.from_numpy(np.ascontiguousarray(c2w))
        self._c2w_host[frame_id] = host
        # kept on the host: the kernels take the camera by value, and a blocking 64-byte upload here waited for everything queued on the
        # stream (1.5 ms per keyframe of hidden sync at the top of every step); get_c2w() uploads on demand
        self.estimated_c2ws[frame_id] = host

    def get_c2w(self, frame_id: int):
        c2w = self.estimated_c2ws.get(frame_id)
        if c2w is not None and c2w.device.type != torch.device(self.device).type:
            c2w = c2w.to(self.device)
        return c2w

    def cam_to_cpu(self, frame_id: int) -> None:
        if frame_id in self.estimated_c2ws:
            self.estimated_c2ws[frame_id] = self.estimated_c2ws[frame_id].cpu()

    def _host_pose(self, frame_id, c2w) -> torch.Tensor:
        host = self._c2w_host.get(frame_id)
        return host if host is not None else c2w.detach().cpu()

    def map(self, frame_data: List[Any], c2w: torch.Tensor) -> None:
        """Reference: vanilla_mapper.py:46-85."""
        self.map_launch(frame_data, c2w)
        self.settle()

    def map_launch(self, frame_data: List[Any], c2w: torch.Tensor, stream=None, defer: bool = False):
        """`map` without the host round trip (MI355X extension): queues the frame's passes; the map's size advances on the device.
        `stream`: a torch stream to queue on instead of the current one (every call of one mapper must use the same stream: the
        passes of consecutive frames depend on each other through the device-resident state).  `defer`: build and RETURN the
        `ovo_map_step_t` (None: nothing to do for this frame) without launching -- the caller hands a round of them to `ovo_round_chain`."""
        frame_id, image, depth_in = frame_data[0], frame_data[1], frame_data[2]
        h, w = depth_in.shape
        near, far = G.depth_range(depth_in)
        if not far > 0:                                    # no valid depth: :60 returns early on a non-empty map, and an empty
            return None                                    # map would receive no point either
        lib = L.load()
        dev = self.device
        depth = G.to_device(depth_in, torch.float32, dev)
        rgb = G.to_device(image, torch.uint8, dev)
        pose = self._host_pose(frame_id, c2w).float().contiguous()
        ds = self.downscale
        n_sub = ((h + ds - 1) // ds) * ((w + ds - 1) // ds)
        if self._n_upper + n_sub > self._cap:              # growing copies `_n` rows: needs the exact size
            if self._deferred:                             # un-launched steps hold the old buffers' addresses and `settle` would wait for them forever
                raise L.OvoHipError("map_launch(defer=True): the map must grow with deferred steps outstanding -- call reserve_round() "
                                    "for the whole round before building its first step")
            self.settle()
            self._reserve(self._n_known + n_sub)
            if stream is not None:                         # the copies into the new buffers were queued on the current stream
                stream.wait_stream(torch.cuda.current_stream())
        self.reap()
        if len(self._pending) >= self._ring.slots - 1:
            if self._deferred:
                raise L.OvoHipError("map_launch(defer=True): result ring full with deferred steps outstanding (size it with ring_slots())")
            self.settle()
        nb = lib.ovo_compact_workspace_bytes(n_sub) + 8
        # built-but-unlaunched steps and the previous round on the chain stream hold these buffers' raw addresses: grow geometrically, and every
        # step keeps the tensors it points at alive through `_keep` below (a dropped block could be handed to another tensor while still in use)
        if self._ws is None or self._ws.numel() < nb:
            self._ws = torch.empty(max(nb, 2 * (self._ws.numel() if self._ws is not None else 0)), dtype=torch.uint8, device=dev)
        ws = self._ws
        if self._explained is None or self._explained.numel() < h * w:
            self._explained = torch.empty(h * w, dtype=torch.uint8, device=dev)
        seq, slot = self._ring.next()
        a = L.MapStep()
        a.map = self.map_ref()
        a.depth, a.rgb, a.h, a.w = depth.data_ptr(), rgb.data_ptr(), h, w
        a.cam = G.frame_camera(near, far, h, w, pose, self._K_host, self.match_distance_th)
        a.K = self._K9
        a.c2w[:] = pose.reshape(-1).tolist()
        a.ds, a.erode, a.n_upper = ds, int(self.k_pooling > 1), self._n_upper
        a.explained, a.ws, a.ws_bytes = self._explained.data_ptr(), ws.data_ptr(), nb
        a.result_host, a.seq = slot, seq
        self._keep.append((depth, rgb, ws, self._explained))   # raw pointers cross the ABI (later, when deferred): keep what the step points at alive
        self._pending.append((seq, n_sub))
        self.last_seq = seq
        if defer:
            self._deferred += 1
            return a
        L.check(lib.ovo_map_step(L.C.byref(a), L.stream() if stream is None else L.C.c_void_p(stream.cuda_stream)))
        return None

    # ------------------------------------------------------------------ map access
    def get_map(self) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """(pcd f32[N,3], pcd_ids i32[N,1], pcd_obj_ids i32[N]) -- views of the live buffers (:98-100)."""
        return self.pcd, self.pcd_ids, self._ins[:self._n]

    def get_kfs(self) -> Dict[int, Dict[str, Any]]:
        return self.kfs

    def update_pcd_obj_ids(self, pcd_objs_ids: torch.Tensor) -> None:
        ids = pcd_objs_ids.reshape(-1)
        if ids.shape[0] != self._n:
            raise ValueError(f"expected {self._n} instance ids, got {ids.shape[0]}")
        if ids.data_ptr() != self._ins.data_ptr():
            self._ins[:self._n].copy_(ids)

    def get_pcd_colors(self) -> np.ndarray:
        return self.pcd_colors.cpu().numpy()

    # ------------------------------------------------------------------ checkpoint format (ovomapping.py:81-116)
    def get_map_dict(self) -> Dict[str, Any]:
        return {"xyz": self.pcd.detach().cpu().clone(), "obj_ids": self.pcd_obj_ids.detach().cpu().clone(),
                "ids": self.pcd_ids.detach().cpu().clone(), "max_id": self.max_id,
                "color": self.pcd_colors.detach().cpu().clone()}

    def set_map_dict(self, map_dict: Dict[str, Any]) -> None:
        n = map_dict["xyz"].shape[0]
        self._n = 0
        self._reserve(max(n, 1))
        self._xyz[:n].copy_(map_dict["xyz"].to(self.device))
        self._ins[:n].copy_(map_dict["obj_ids"].reshape(-1).to(self.device))
        self._ids[:n].copy_(map_dict["ids"].reshape(-1).to(self.device))
        self._rgb[:n].copy_(map_dict["color"].to(self.device))
        self._n, self.max_id = n, map_dict["max_id"]

    def get_cam_dict(self) -> Dict[str, Any]:
        return {k: v.cpu().numpy() for k, v in self.estimated_c2ws.items()}

    def set_cam_dict(self, cam_dict: Dict[str, Any]) -> None:
        self.estimated_c2ws, self._c2w_host = {}, {}
        for k, v in cam_dict.items():
            host = torch.from_numpy(v)
            self._c2w_host[int(k)] = host
            self.estimated_c2ws[int(k)] = host
