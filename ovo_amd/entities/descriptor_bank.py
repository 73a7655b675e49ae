"""Device-resident descriptor storage for the multi-view fusion.

The reference keeps every per-(keyframe, instance) descriptor as a separate CPU tensor in nested dicts,
re-stacks them per instance on every update (instance3d.py:157-189) and re-uploads all instance
descriptors on every query (ovo.py:513-527).  Here:

  * `store`  f32[R, D]   append-only rows, one per (keyframe, instance) descriptor, on the GPU;
  * `table`  f32[S, D]   one fused descriptor per instance slot, on the GPU -- `OVO.query` reads it directly;
  * fusion of any number of instances is ONE launch of `ovo_fuse_views` over a CSR view list.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch

from .. import _lib as L

FUSION_MODES = {"avg_pooling": 0, "l1_medoid": 1, "cossim_medoid": 2}


class KeyframeView:
    """Read-only mapping {ins_id: descriptor} of one keyframe; values are views into the bank's store."""

    def __init__(self, bank: "DescriptorBank", rows: Dict[int, int]):
        self._bank, self._rows = bank, rows

    def __getitem__(self, ins_id: int) -> torch.Tensor:
        return self._bank.store[self._rows[ins_id]]

    def __contains__(self, ins_id) -> bool:
        return ins_id in self._rows

    def __iter__(self):
        return iter(self._rows)

    def __len__(self) -> int:
        return len(self._rows)

    def keys(self):
        return self._rows.keys()

    def items(self):
        return ((k, self[k]) for k in self._rows)

    def row(self, ins_id: int) -> int:
        return self._rows[ins_id]

    def pop(self, ins_id: int) -> torch.Tensor:
        return self._bank.store[self._rows.pop(ins_id)]

    def __setitem__(self, ins_id: int, value) -> None:
        """Re-key a descriptor (ovo.py:417-419 moves id2's descriptor to id1 after an instance merge)."""
        if isinstance(value, torch.Tensor) and value.data_ptr() >= self._bank.store.data_ptr() and value.dim() == 1:
            off = (value.data_ptr() - self._bank.store.data_ptr()) // (4 * self._bank.dim)
            if 0 <= off < self._bank.n_rows:
                self._rows[ins_id] = int(off)
                return
        self._rows[ins_id] = self._bank.append(value.reshape(1, -1))[0]


class _LazyMedoids(dict):
    """slot -> clip_feature_kf.  The medoid fusions pick a view ON THE DEVICE; the index is only needed when a checkpoint is exported
    (Instance3D.clip_feature_kf), so `fuse` parks (tensor, position) here and the device->host copy happens on the first read --
    not once per keyframe on the frame's critical path."""

    def _resolve(self, v):
        if isinstance(v, tuple):
            t, k = v
            cache = getattr(t, "_ovo_host", None)
            if cache is None:
                cache = t.tolist()                      # one copy per fusion launch, shared by every slot it updated
                t._ovo_host = cache
            return int(cache[k])
        return v

    def __getitem__(self, key):
        v = dict.__getitem__(self, key)
        r = self._resolve(v)
        if r is not v:
            dict.__setitem__(self, key, r)
        return r

    def get(self, key, default=None):
        return self[key] if key in self else default


class DescriptorBank:
    def __init__(self, dim: int, device, rows: int = 4096, slots: int = 1024):
        self.dim, self.device = int(dim), device
        self.store = torch.empty((rows, self.dim), dtype=torch.float32, device=device)
        self.table = torch.zeros((slots, self.dim), dtype=torch.float32, device=device)
        self.sums: Optional[torch.Tensor] = None   # running sums of avg_pooling (fuse_add), same shape as table, allocated on first use
        self.n_rows = 0
        self.slot_of: Dict[int, int] = {}
        self.views_of: Dict[int, int] = {}       # slot -> number of views used by the last fusion
        self.medoid_of: "_LazyMedoids" = _LazyMedoids()    # slot -> clip_feature_kf (position of the picked view), resolved on read

    # ---------------------------------------------------------------- storage
    def append(self, feats: torch.Tensor) -> List[int]:
        """Append f32[n, D] descriptors (device) and return their row numbers."""
        n = feats.shape[0]
        if self.n_rows + n > self.store.shape[0]:
            grown = torch.empty((max(2 * self.store.shape[0], self.n_rows + n), self.dim), dtype=torch.float32, device=self.device)
            grown[:self.n_rows].copy_(self.store[:self.n_rows])
            self.store = grown
        self.store[self.n_rows:self.n_rows + n].copy_(feats.to(device=self.device, dtype=torch.float32))
        rows = list(range(self.n_rows, self.n_rows + n))
        self.n_rows += n
        return rows

    def slot(self, ins_id: int) -> int:
        s = self.slot_of.get(ins_id)
        if s is None:
            s = len(self.slot_of)
            if s >= self.table.shape[0]:
                grown = torch.zeros((2 * self.table.shape[0], self.dim), dtype=torch.float32, device=self.device)
                grown[:self.table.shape[0]].copy_(self.table)
                if self.sums is not None:
                    g2 = torch.zeros_like(grown)
                    g2[:self.table.shape[0]].copy_(self.sums)
                    self.sums = g2
                self.table = grown
            self.slot_of[ins_id] = s
        return s

    def has_feature(self, ins_id: int) -> bool:
        s = self.slot_of.get(ins_id)
        return s is not None and s in self.views_of

    def feature(self, ins_id: int) -> torch.Tensor:
        """Fused descriptor with the reference's shape: [D] after a single view, [1, D] after a fusion."""
        s = self.slot_of[ins_id]
        row = self.table[s]
        return row if self.views_of[s] == 1 else row[None]

    def set_feature(self, ins_id: int, value: torch.Tensor) -> None:
        s = self.slot(ins_id)
        self.table[s].copy_(value.reshape(-1).to(device=self.device, dtype=torch.float32))
        self.views_of[s] = 1 if value.dim() == 1 else 2
        self.medoid_of[s] = None

    # ---------------------------------------------------------------- fusion
    def fuse(self, updates: Sequence[Tuple[int, Sequence[int]]], mode: str) -> Dict[int, Optional[int]]:
        """updates: (ins_id, store rows in the reference's stacking order).  One kernel launch.

        Returns {ins_id: clip_feature_kf} = 0 for one view, None for avg_pooling, medoid position otherwise."""
        updates = [(i, list(r)) for i, r in updates if len(r) > 0]
        if not updates:
            return {}
        if mode not in FUSION_MODES:
            raise NotImplementedError(f"fusion '{mode}' (camfusion needs a checkpoint the reference does not ship)")
        off, rows, slots = [0], [], []
        for ins_id, r in updates:
            rows.extend(r)
            off.append(len(rows))
            slots.append(self.slot(ins_id))
        n = len(updates)
        meta = torch.tensor(off + rows + slots, dtype=torch.int32).to(self.device, non_blocking=True)
        csr_off, csr_rows, t_rows = meta[:n + 1], meta[n + 1:n + 1 + len(rows)], meta[n + 1 + len(rows):]
        m = FUSION_MODES[mode]
        out_view = torch.empty(n, dtype=torch.int32, device=self.device) if m else None
        L.check(L.load().ovo_fuse_views(L.ptr(self.store), self.dim, L.ptr(csr_off), L.ptr(csr_rows), n, m,
                                        L.ptr(self.table), L.ptr(t_rows), L.ptr(out_view), L.stream()))
        result = _LazyMedoids()
        for k, (ins_id, r) in enumerate(updates):
            s = slots[k]
            self.views_of[s] = len(r)
            kf = 0 if len(r) == 1 else (None if m == 0 else (out_view, k))     # medoid position: resolved lazily (no sync here)
            dict.__setitem__(self.medoid_of, s, kf)
            dict.__setitem__(result, ins_id, kf)
        return result

    def fuse_add(self, updates: Sequence[Tuple[int, Sequence[int], int]]) -> None:
        """avg_pooling as a running sum (`ovo_fuse_views_add`).  updates: (ins_id, store rows of the NEW views, views already in the sum --
        0 starts the sum over).  One kernel launch; the CSR holds only the new rows."""
        updates = [(i, list(r), int(b)) for i, r, b in updates if len(r) > 0]
        if not updates:
            return
        off, rows, slots, before = [0], [], [], []
        for ins_id, r, b in updates:
            rows.extend(r)
            off.append(len(rows))
            slots.append(self.slot(ins_id))
            before.append(b)
        if self.sums is None or self.sums.shape[0] < self.table.shape[0]:
            grown = torch.zeros_like(self.table)
            if self.sums is not None:
                grown[:self.sums.shape[0]].copy_(self.sums)
            self.sums = grown
        n = len(updates)
        meta = torch.tensor(off + rows + slots + before, dtype=torch.int32).to(self.device, non_blocking=True)
        csr_off, csr_rows = meta[:n + 1], meta[n + 1:n + 1 + len(rows)]
        t_rows, d_before = meta[n + 1 + len(rows):n + 1 + len(rows) + n], meta[n + 1 + len(rows) + n:]
        L.check(L.load().ovo_fuse_views_add(L.ptr(self.store), self.dim, L.ptr(csr_off), L.ptr(csr_rows), L.ptr(d_before), n, L.ptr(self.sums),
                                            L.ptr(self.table), L.ptr(t_rows), L.stream()))
        for (ins_id, r, b), s in zip(updates, slots):
            self.views_of[s] = b + len(r)
            dict.__setitem__(self.medoid_of, s, 0 if b + len(r) == 1 else None)

    def gather(self, ins_ids: Iterable[int]) -> torch.Tensor:
        """f32[N, D] fused descriptors in the given order (the resident replacement of ovo.py:513-527)."""
        slots = [self.slot_of[i] for i in ins_ids]
        if (self.dim * 4) % 16 == 0:
            return L.gather_rows(self.table, slots)
        return self.table.index_select(0, torch.tensor(slots, dtype=torch.int64).to(self.device))
