"""One launch for the map + tracking chains of a whole round of keyframes (`ovo_round_chain`, csrc/geometry.hip:k_round_chain).

`VanillaMapper.map_launch(defer=True)` and `OVO.detect_and_track_launch(defer=True)` build a keyframe's `ovo_map_step_t` /
`ovo_track_step_t` without launching; `RoundLauncher.launch` hands a round of them to the library in one call.  What the reference does
per keyframe with a `.sum()` for the map size and >= 3 `.item()` round trips per mask (vanilla_mapper.py:81-85, ovo.py:255-282) is then one
kernel launch per ROUND and one pinned result block per keyframe."""
from __future__ import annotations

import os
from typing import List, Optional

import torch

from .. import _lib as L


class RoundLauncher:
    def __init__(self, device, workgroups: Optional[int] = None):
        self.device = device
        # Measured (profiles/r03_round_emulation.txt): with 64 / 128 persistent workgroups the passes run at one wave per SIMD and are
        # latency-bound -- an emulated 8-rank round takes 6.0 / 5.5 ms against 4.65 ms for the per-pass launches on their own stream --
        # so the one-launch form is opt-in (OVO_ROUND_CHAIN=1) until its passes keep more loads in flight.
        self.enabled = bool(os.environ.get("OVO_ROUND_CHAIN")) if workgroups is None else True
        self.workgroups = int(os.environ.get("OVO_CHAIN_WORKGROUPS", "0")) if workgroups is None else int(workgroups)
        self.merged = not os.environ.get("OVO_NO_KEYFRAME_STEP")
        self._ctx = None
        self.launches = 0                # rounds that went through ovo_round_chain
        self.fallbacks = 0               # rounds that went keyframe by keyframe

    def launch(self, maps: List["L.MapStep"], tracks: List["L.TrackStep"], stream=None) -> None:
        """maps[k] / tracks[k]: the two steps of keyframe k (`depth == NULL` / `n_masks == 0`: that half is absent)."""
        lib = L.load()
        n = len(maps)
        if n == 0:
            return
        handle = L.stream() if stream is None else L.C.c_void_p(stream.cuda_stream)
        rc = L.E_UNSUPPORTED
        if self.enabled and lib.ovo_round_chain_params_bytes() == 0:    # a production build carries no one-launch form (--experimental does)
            self.enabled = False
        if self.enabled:
            if self._ctx is None:
                self._bar = torch.zeros(2, dtype=torch.int64, device=self.device)
                self._params = lib.ovo_host_alloc(lib.ovo_round_chain_params_bytes())
                if not self._params:
                    raise L.OvoHipError(lib.ovo_hip_last_error().decode())
                self._ctx = L.RoundChain(self._params, self._bar.data_ptr(), 0, 0, self.workgroups)
            rc = lib.ovo_round_chain(L.C.byref(self._ctx), (L.MapStep * n)(*maps), (L.TrackStep * n)(*tracks), n, handle)
        if rc == L.E_UNSUPPORTED:                                  # shapes the one-launch form does not cover: the two calls per keyframe
            self.fallbacks += 1
            for m, t in zip(maps, tracks):
                if m.depth and t.n_masks > 0 and self.merged:         # both halves: their independent passes share launches
                    L.check(lib.ovo_keyframe_step(L.C.byref(m), L.C.byref(t), handle))
                    continue
                if m.depth:
                    L.check(lib.ovo_map_step(L.C.byref(m), handle))
                if t.n_masks > 0:
                    L.check(lib.ovo_track_step(L.C.byref(t), handle))
            return
        L.check(rc)
        self.launches += 1

    def __del__(self):
        try:
            if getattr(self, "_params", None):
                L.load().ovo_host_free(self._params)
        except Exception:
            pass
