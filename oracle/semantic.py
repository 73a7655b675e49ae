"""Oracle: point-map building, mask<->3D-instance tracking and per-instance bookkeeping (numpy).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Plain loops, no device code.
"""
from __future__ import annotations

import heapq
from typing import Dict, List, Tuple

import numpy as np
import torch

from . import geometry as G


# ---------------------------------------------------------------------------------------------
class InstanceRecord:
    """State of one 3D instance -- instance3d.py:28-155 (update / add_top_kf / _add_top_kf)."""

    def __init__(self, ins_id: int, n_top: int):
        self.id = ins_id
        self.n_top = n_top
        self.kfs: List[int] = []
        self.points: List[int] = []
        self.heap: List[Tuple[int, int]] = []      # (area, kf) min-heap
        self.dirty = False
        self.feature = None
        self.feature_kf = None

    def _slot(self, kf):
        for i, (_, k) in enumerate(self.heap):
            if k == kf:
                return i
        return -1

    def in_top(self, kf) -> bool:
        return self._slot(kf) > -1

    def offer_view(self, kf: int, area: int) -> None:
        """instance3d.py:105-137."""
        i = self._slot(kf)
        if i > -1:
            if area > self.heap[i][0]:
                self.heap[i] = (area, kf)
                heapq.heapify(self.heap)
                self.dirty = True
            return
        if len(self.heap) < self.n_top:
            heapq.heappush(self.heap, (area, kf))
            self.dirty = True
        else:
            dropped = heapq.heappushpop(self.heap, (area, kf))
            if self.n_top <= 0 or dropped[1] != kf:
                self.dirty = True

    def observe(self, point_ids: List[int], kf: int, area: int) -> None:
        """instance3d.py:77-103."""
        if kf not in self.kfs:
            self.kfs.append(kf)
        self.points.extend(point_ids)
        self.offer_view(kf, area)

    def refresh_feature(self, kf_features: Dict[int, Dict[int, np.ndarray]], fusion: str, force=False) -> None:
        """instance3d.py:157-189 update_clip."""
        if not (self.dirty or force):
            return
        if self.n_top > 0:
            views = [kf for _, kf in heapq.nlargest(self.n_top, self.heap)]
        else:
            views = list(self.kfs)
        rows = [kf_features[kf][self.id] for kf in views if kf_features.get(kf) is not None]
        if not rows:
            return
        rows = np.stack(rows).astype(np.float32)
        if rows.shape[0] == 1:
            self.feature, self.feature_kf = rows[0], 0
        else:
            self.feature, self.feature_kf = fuse_views(rows, fusion)
        self.dirty = False


def fuse_views(rows: np.ndarray, fusion: str):
    """instance3d.py:9-21 l1_medoid / cossim_medoid / avg_pooling on rows [V, D]."""
    if fusion == "avg_pooling":
        return torch.from_numpy(rows).mean(dim=-2).numpy(), None
    if fusion == "l1_medoid":
        d = np.abs(rows[:, None, :] - rows[None, :, :]).sum(-1).sum(0)
        k = int(d.argmin())
        return rows[k], k
    if fusion == "cossim_medoid":
        t = torch.from_numpy(rows)[None]
        s = torch.cosine_similarity(t, t.permute(1, 0, 2), dim=-1).sum(-1)
        k = int(s.argmax())
        return rows[k], k
    raise NotImplementedError(fusion)


# ---------------------------------------------------------------------------------------------
class PointMap:
    """vanilla_mapper.py:19-136 VanillaMapper state + map()."""

    def __init__(self, K: np.ndarray, k_pooling: int = 3, downscale: int = 2):
        self.K = np.asarray(K, np.float32)
        self.xyz = np.zeros((0, 3), np.float32)
        self.ids = np.zeros((0, 1), np.int32)
        self.ins = np.zeros((0,), np.int32)
        self.rgb = np.zeros((0, 3), np.uint8)
        self.next_id = 0
        self.k_pooling, self.ds = k_pooling, downscale
        self.th = 0.03          # vanilla_mapper.py:17

    def integrate(self, rgb: np.ndarray, depth: np.ndarray, c2w: np.ndarray) -> int:
        depth = depth.astype(np.float32)
        explained = None
        if self.next_id > 0:
            corners = G.frustum_corners(depth, c2w, self.K)
            fids = G.frustum_point_ids(self.xyz, corners)
            w2c = torch.linalg.inv(torch.from_numpy(np.asarray(c2w, np.float32))).numpy()
            _, uv = G.match(depth, w2c, self.xyz[fids], self.K, self.th)
            explained = np.zeros(depth.shape, np.uint8)
            explained[uv[:, 1], uv[:, 0]] = 1
        xyz, col = G.backproject(depth, rgb, explained, self.K, c2w,
                                 erode=self.next_id > 0 and self.k_pooling > 1, ds=self.ds)
        m = xyz.shape[0]
        if m == 0:
            return 0
        self.xyz = np.vstack([self.xyz, xyz])
        self.ids = np.vstack([self.ids, np.arange(self.next_id, self.next_id + m, dtype=np.int32)[:, None]])
        self.ins = np.concatenate([self.ins, np.full(m, -1, np.int32)])
        self.rgb = np.vstack([self.rgb, col])
        self.next_id += m
        return m


# ---------------------------------------------------------------------------------------------
def smallest_mode(values: np.ndarray) -> int:
    """torch.mode on CPU: most frequent value, smallest among ties."""
    u, c = np.unique(values, return_counts=True)
    return int(u[c == c.max()].min())


class SemanticTracker:
    """ovo.py:182-324: _match_and_track_instances / _track_objects / _fuse_masks_with_same_ins_id."""

    def __init__(self, K, match_th=0.05, track_th=100, depth_filter=False, n_top=0, fusion="avg_pooling"):
        self.K = np.asarray(K, np.float32)
        self.match_th, self.track_th, self.depth_filter = match_th, track_th, depth_filter
        self.n_top, self.fusion = n_top, fusion
        self.objects: Dict[int, InstanceRecord] = {}
        self.next_ins = 0
        self.kf = 0
        self.kf_features: Dict[int, Dict[int, np.ndarray]] = {}

    def step(self, depth, ratio, pts, pt_ids, pt_ins, c2w, seg_map, masks):
        """-> (matched_ins_ids, fused masks bool[M,H,W], n_matched, updated i32[N])."""
        depth = depth.astype(np.float32)
        corners = G.frustum_corners(depth, c2w, self.K)
        fids = G.frustum_point_ids(pts, corners)
        if self.depth_filter:
            depth = G.depth_filter(depth)
        w2c = torch.linalg.inv(torch.from_numpy(np.asarray(c2w, np.float32))).numpy()
        midx, uv = G.match(depth, w2c, pts[fids], self.K, self.match_th)
        if len(ratio) > 0:                                        # ovo.py:218-221
            uv = uv + np.int32(ratio[-1])
            v = (uv[:, 1].astype(np.float32) * np.float32(ratio[0])).astype(np.int32)
            u = (uv[:, 0].astype(np.float32) * np.float32(ratio[1])).astype(np.int32)
            uv = np.stack([u, v], 1)
        seg = seg_map[uv[:, 1], uv[:, 0]]
        f_ids = np.asarray(pt_ids).reshape(-1)[fids]
        f_ins = np.asarray(pt_ins, np.int32)[fids].copy()
        info: Dict[int, List[Tuple[int, int]]] = {}
        for m in range(int(seg_map.max()) + 1):                    # ovo.py:255-280
            target = -1
            mp = midx[seg == m]
            if mp.shape[0] <= self.track_th:
                continue
            area = int((seg_map == m).sum())
            assigned = f_ins[mp] > -1
            fresh = [int(i) for i in f_ids[mp[~assigned]]]
            if int(assigned.sum()) > self.track_th:
                target = smallest_mode(f_ins[mp[assigned]])
                self.objects[target].observe(fresh, self.kf, area)
                info.setdefault(target, []).append((m, area))
            elif len(fresh) > self.track_th:
                target = self.next_ins
                self.next_ins += 1
                rec = InstanceRecord(target, self.n_top)
                rec.observe(fresh, self.kf, area)
                self.objects[target] = rec
                info[target] = [(m, area)]
            if target > -1:
                f_ins[mp[~assigned]] = target
        masks = masks.copy()
        matched, rows = [], []
        for ins_id, hits in list(info.items()):                   # ovo.py:299-322
            first = hits[0][0]
            if len(hits) > 1:
                for other, _ in hits[1:]:
                    masks[first] |= masks[other]
                if self.n_top > 0:
                    self.objects[ins_id].offer_view(self.kf, int(masks[first].sum()))
            if self.n_top <= 0 or self.objects[ins_id].in_top(self.kf):
                matched.append(ins_id)
                rows.append(first)
        updated = np.asarray(pt_ins, np.int32).copy()
        updated[fids] = f_ins
        self.kf += 1
        return matched, masks[rows], int(midx.shape[0]), updated

    def add_features(self, kf: int, matched: List[int], feats: np.ndarray) -> None:
        """ovo.py:440-461 _update_matched_objects_clip."""
        self.kf_features[kf] = {i: feats[j] for j, i in enumerate(matched) if i != -1}
        for i in matched:
            self.objects[i].refresh_feature(self.kf_features, self.fusion)

    def feature_table(self) -> np.ndarray:
        """ovo.py:513-527 get_objs_clips: rows follow dict insertion order of self.objects."""
        return np.stack([o.feature for o in self.objects.values()]).astype(np.float32)


def merge_instances(xyz: np.ndarray, ins: np.ndarray, ids, feats, th_centroid: float = 1.5, th_cossim: float = 0.81, th_points: float = 0.1,
                    same=None):
    """ovo.py:381-407 (update_map steps 1-2) with instance_utils.py:5-35, literally: instances without map points are
    dropped, then every ordered pair is tested (centroid distance, descriptor cosine, fraction of points of the first
    whose nearest neighbour in the second is closer than th_points) and merged greedily in dictionary order.
    Open3D's `compute_point_cloud_distance` (a KD-tree nearest-neighbour query in double precision) is restated with
    scipy's cKDTree -- Open3D is not installed here; the rest is pinned by tests/golden/loopclose.npz (the reference's own
    update_map run with that same stand-in, and with `same_instance` replaced by a table: `same(id1, id2)` here).
    ids: instance ids in `objects` order; feats: {id: f32[D]}.  -> (kept ids, {merged id: surviving id}, updated ins)."""
    import torch
    from scipy.spatial import cKDTree
    ins = ins.copy()
    present = set(np.unique(ins).tolist())
    objects_list = [i for i in ids if i in present]
    pcds = {i: xyz[ins == i] for i in objects_list}
    cents = {i: torch.from_numpy(pcds[i]).mean(axis=0) for i in objects_list}

    def geometric(i1, i2):
        if ((cents[i1] - cents[i2]) ** 2).sum().sqrt() > th_centroid:
            return False
        cos = torch.nn.functional.cosine_similarity(torch.from_numpy(feats[i1]), torch.from_numpy(feats[i2]), dim=0)
        if cos < th_cossim:
            return False
        d, _ = cKDTree(pcds[i2].astype(np.float64)).query(pcds[i1].astype(np.float64))
        p = (d < th_points).astype(float).mean()
        return bool(p > 0.5 or (cos > 0.9 and p > 0.2))
    same = same or geometric
    kept, fused = [], {}
    for a, i1 in enumerate(objects_list):
        if i1 in fused:
            continue
        for i2 in objects_list[a + 1:]:
            if i2 in fused:
                continue
            if same(i1, i2):
                ins[ins == i2] = i1
                fused[i2] = i1
        kept.append(i1)
    return kept, fused, ins
